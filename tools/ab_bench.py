#!/usr/bin/env python3
"""A/B kernel timing in ONE process (interleaved rounds): tools/ab_bench.py --option v3_rw --variants 1 4 --rounds 5
Reports the blind-rotate and keyswitch kernel durations (HIP events) per variant, median and min, and
checks every variant's output against variant-independent expectations (decrypts to NAND; all equal)."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe

ap = argparse.ArgumentParser()
ap.add_argument("--variants", type=int, nargs="+", default=[1, 2])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--gates", type=int, default=4096)
ap.add_argument("--params", default="80")
ap.add_argument("--option", default="v3_rw")
args = ap.parse_args()

params = tfhe.tfhe_parameters_80() if args.params == "80" else tfhe.tfhe_parameters_128()
rng = np.random.default_rng(123)
sk, ck = tfhe.make_key_pair(rng, params)
eng = ck.engine(0)
B = args.gates
bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
x, y = tfhe.encrypt(rng, sk, bx).data, tfhe.encrypt(rng, sk, by).data
ops = np.zeros(B, np.uint8)
res = {v: {"br": [], "ks": []} for v in args.variants}
ref = None
for r in range(args.rounds + 1):
    for v in args.variants:
        eng.set_option(args.option, v)
        out = eng.gates(ops, x, y)
        if r == 0:
            assert np.array_equal(tfhe.decrypt(sk, out), ~(bx & by)), f"variant {v}: wrong results"
            if ref is None: ref = out
            assert np.array_equal(out, ref), f"variant {v}: differs from variant {args.variants[0]}"
            continue  # warm-up round
        res[v]["br"].append(eng.last_timing_ms(0))
        res[v]["ks"].append(eng.last_timing_ms(1))
for v in args.variants:
    br, ks = np.array(res[v]["br"]), np.array(res[v]["ks"])
    print(f"{args.option}={v}: BR median {np.median(br):.3f} ms min {br.min():.3f}  |  KS median {np.median(ks):.3f} ms min {ks.min():.3f}"
          f"  |  {B / np.median(br) * 1e3:.0f} rot/s  frac_hbm {B / np.median(br) * 1e3 * 16384000 / 8e12:.3f}")
