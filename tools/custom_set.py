#!/usr/bin/env python3
"""A parameter set no tuned instantiation was written for: blind-rotate time of B NAND gates on the kernels the dispatcher picks
(N = 1024, k = 1, any l: the run-time-l instantiations; any other N or k: the general / any-N kernels) and, where it exists, on the
general kernel (option br_general) — or with --anyn on the any-N kernel (option br_anyn) beside the default — with decrypt check.
  python tools/custom_set.py --l 4 --beta 8 [--n 500] [--N 1024] [--k 1] [--gates 4096] [--anyn]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import tfhe_jl_amd as tfhe

ap = argparse.ArgumentParser()
ap.add_argument("--l", type=int, required=True); ap.add_argument("--beta", type=int, required=True)
ap.add_argument("--n", type=int, default=500); ap.add_argument("--gates", type=int, default=4096); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--N", type=int, default=1024); ap.add_argument("--k", type=int, default=1); ap.add_argument("--anyn", action="store_true")
a = ap.parse_args()
p = tfhe.SchemeParameters(a.n, 1 / 2**15, a.N, a.k, a.l, a.beta, 9e-9, 8, 2, 1 / 2**15, 1)
rng = np.random.default_rng(1)
sk, ck = tfhe.make_key_pair(rng, p, keygen="device")
eng = ck.engine(0)
bits = rng.integers(0, 2, (2, a.gates)).astype(bool)
x, y = (tfhe.encrypt(rng, sk, b).data for b in bits)
ops = np.zeros(a.gates, np.uint8)
res = {"N": a.N, "k": a.k, "l": a.l, "beta": a.beta, "n": a.n, "gates": a.gates}
if a.anyn:
    e2 = tfhe.Engine(p, 0)
    e2.set_option("br_anyn", 1)
    e2.load_bootstrap_key(ck.bootstrap_key)
    e2.load_keyswitch_key(ck.keyswitch_key)
engines = [(eng, 0, "default")] + ([(e2, 0, "anyn")] if a.anyn else [(eng, 1, "general")] if a.N in (1024, 2048) and a.k <= 4 else [])
for eng, general, label in engines:
    eng.set_option("br_general", general)
    out = eng.gates(ops, x, y)
    t = []
    for _ in range(a.reps):
        eng.gates(ops, x, y); t.append(eng.last_timing_ms(0))
    res[label] = {"kernel": eng.last_kernel_name(), "blind_rotate_ms": float(np.median(t)), "keyswitch_ms": eng.last_timing_ms(1),
                                                "decrypt_ok": float((tfhe.decrypt(sk, out) == ~(bits[0] & bits[1])).mean())}
print(json.dumps(res))
