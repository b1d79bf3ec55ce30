#!/usr/bin/env python3
"""A parameter set no tuned instantiation was written for (k = 1, N = 1024, any l): blind-rotate time of B NAND gates on the kernels
the dispatcher picks (the run-time-l instantiations) and on the general kernel (option br_general), with decrypt check.
  python tools/custom_set.py --l 4 --beta 8 [--n 500] [--gates 4096]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import tfhe_jl_amd as tfhe

ap = argparse.ArgumentParser()
ap.add_argument("--l", type=int, required=True); ap.add_argument("--beta", type=int, required=True)
ap.add_argument("--n", type=int, default=500); ap.add_argument("--gates", type=int, default=4096); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
p = tfhe.SchemeParameters(a.n, 1 / 2**15, 1024, 1, a.l, a.beta, 9e-9, 8, 2, 1 / 2**15, 1)
rng = np.random.default_rng(1)
sk, ck = tfhe.make_key_pair(rng, p, keygen="device")
eng = ck.engine(0)
bits = rng.integers(0, 2, (2, a.gates)).astype(bool)
x, y = (tfhe.encrypt(rng, sk, b).data for b in bits)
ops = np.zeros(a.gates, np.uint8)
res = {"l": a.l, "beta": a.beta, "n": a.n, "gates": a.gates}
for general in (0, 1):
    eng.set_option("br_general", general)
    out = eng.gates(ops, x, y)
    t = []
    for _ in range(a.reps):
        eng.gates(ops, x, y); t.append(eng.last_timing_ms(0))
    res["general" if general else "default"] = {"kernel": eng.last_kernel_name(), "blind_rotate_ms": float(np.median(t)),
                                                "decrypt_ok": float((tfhe.decrypt(sk, out) == ~(bits[0] & bits[1])).mean())}
print(json.dumps(res))
