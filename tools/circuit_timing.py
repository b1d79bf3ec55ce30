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
