import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import tfhe_jl_amd as tfhe
rng = np.random.default_rng(77)
sk, ck = tfhe.make_key_pair(rng, tfhe.tfhe_parameters_80(tlwe_mask_size=2))
e = ck.engine(0)
B = 4096
bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
x, y = tfhe.encrypt(rng, sk, bx).data, tfhe.encrypt(rng, sk, by).data
ops = np.zeros(B, np.uint8)
out = e.gates(ops, x, y)
br = []
for _ in range(3):
    e.gates(ops, x, y); br.append(e.last_timing_ms(0))
print(f"k=2 B={B}: blind rotate {np.median(br):.2f} ms ({B/np.median(br)*1e3:.0f} rot/s, frac {B/np.median(br)*1e3*500*2*9*1024*4/8e12:.3f}), ks {e.last_timing_ms(1):.2f} ms, decrypt ok {float((tfhe.decrypt(sk,out)==~(bx&by)).mean())}")
