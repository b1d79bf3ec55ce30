"""Host time inside the raw tfhe_gates_level ccall per circuit level (16-MUX levels back to back): what showed that the call blocks
only once the host is four levels ahead (the ring of pinned staging blocks)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tfhe_jl_amd as tfhe
rng = np.random.default_rng(5)
p = tfhe.tfhe_parameters_80()
sk, ck = tfhe.make_key_pair(rng, p, keygen="device")
W = 16
enc = tfhe.encrypt(rng, sk, rng.integers(0, 2, 3 * W).astype(bool)).data
eng = ck.engine(0)
eng.set_option("timing_events", 0)
eng.wires_alloc(5 * W)
eng.wires_upload(0, enc)
ops = np.full(W, tfhe.OPCODES["MUX"], np.uint8)
sel = np.arange(W, dtype=np.int32); third = np.arange(2 * W, 3 * W, dtype=np.int32)
banks = [np.arange(3 * W, 4 * W, dtype=np.int32), np.arange(4 * W, 5 * W, dtype=np.int32)]
prev = np.arange(W, 2 * W, dtype=np.int32)
lib = eng._lib
from tfhe_jl_amd._lib import _ptr
for t in range(40):
    out = banks[t & 1]
    b = np.ascontiguousarray(prev[::-1])
    t0 = time.perf_counter()
    rc = lib.tfhe_gates_level(eng._h, _ptr(ops), _ptr(sel), _ptr(b), _ptr(third), _ptr(out), W)
    t1 = time.perf_counter()
    if t >= 30: print(f"level {t}: ccall {1e6*(t1-t0):.0f} us rc={rc}")
    prev = out
t0 = time.perf_counter(); eng.wires_gather(prev); print("final gather", 1e6*(time.perf_counter()-t0))
