import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe
rng = np.random.default_rng(2048)
p = tfhe.SchemeParameters(630, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)
sk, ck = tfhe.make_key_pair(rng, p)
e = ck.engine(0)
B = 4096
bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
x, y = tfhe.encrypt(rng, sk, bx).data, tfhe.encrypt(rng, sk, by).data
ops = np.zeros(B, np.uint8)
out = e.gates(ops, x, y)
br = []
for _ in range(3):
    e.gates(ops, x, y); br.append(e.last_timing_ms(0))
print(f"N=2048 B={B}: blind rotate {np.median(br):.2f} ms ({B/np.median(br)*1e3:.0f} rot/s, frac {B/np.median(br)*1e3*61931520/8e12:.3f}), ks {e.last_timing_ms(1):.2f} ms, decrypt ok {float((tfhe.decrypt(sk,out)==~(bx&by)).mean())}")
