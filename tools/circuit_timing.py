import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "examples"))
import tfhe_jl_amd as tfhe
from tutorial import prepare, encrypted_minimum_circuit, bits_to_int
sk, ck, c1, c2 = prepare()
circ = encrypted_minimum_circuit(16)
inputs = tfhe.LweSampleArray(np.concatenate([c1.data, c2.data]))
for it in range(3):
    t0 = time.perf_counter(); out = circ.run(ck, inputs); dt = time.perf_counter() - t0
    print(f"run {it}: {dt*1e3:.1f} ms -> {bits_to_int(tfhe.decrypt(sk, out))}")
eng = ck.engine(0)
# per-level timing
eng.wires_alloc(circ.num_wires); eng.wires_upload(0, inputs.data)
t0 = time.perf_counter()
for ops, a, b, c, o in circ.level_arrays():
    eng.gates_level(ops, a, b, c, o)
eng.wires_download(0, 1)
print(f"18 levels device time + launches: {(time.perf_counter()-t0)*1e3:.1f} ms")
tree = encrypted_minimum_circuit(16, log_depth=True)
for it in range(3):
    t0 = time.perf_counter(); out = tree.run(ck, inputs); dt = time.perf_counter() - t0
    print(f"log-depth variant ({len(tree.levels())} levels, {sum(len(l) for l in tree.levels())} gates) run {it}: {dt*1e3:.1f} ms -> {bits_to_int(tfhe.decrypt(sk, out))}")
# the same circuit on many input sets at once (Circuit.run_batch): every level is one call over all instances
for M in (64, 128, 1024):
    many = np.broadcast_to(inputs.data, (M,) + inputs.data.shape).copy()
    circ.run_batch(ck, many[:2])
    t0 = time.perf_counter(); res = circ.run_batch(ck, many); dt = time.perf_counter() - t0
    ok = all(bits_to_int(tfhe.decrypt(sk, tfhe.LweSampleArray(res[i]))) == 42 for i in (0, M // 2, M - 1))
    print(f"{M} instances of the 18-level circuit in one run_batch: {dt*1e3:.1f} ms = {dt*1e3/M:.3f} ms per instance ({'ok' if ok else 'WRONG'})")
