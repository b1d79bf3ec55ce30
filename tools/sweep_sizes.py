#!/usr/bin/env python3
"""Blind-rotate kernel time of the engine's dispatch at given batch sizes, with and without engine options (one process,
one device: an interleaved A/B).
  python tools/sweep_sizes.py --params 80|128|k2|2048 --sizes 1100,3072 --ab br_split=0 [--ab k2_rw=3] [--reps 5]
Every --ab NAME=VALUE is one alternative measured beside the defaults."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe
ap = argparse.ArgumentParser()
ap.add_argument("--params", default="80")
ap.add_argument("--sizes", default="1100,3072")
ap.add_argument("--ab", action="append", default=[])
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
rng = np.random.default_rng(123)
P = {"80": tfhe.tfhe_parameters_80(), "128": tfhe.tfhe_parameters_128(), "k2": tfhe.tfhe_parameters_80(tlwe_mask_size=2),
     "2048": tfhe.SchemeParameters(630, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)}[a.params]      # "2048": BASELINE config 4b's synthetic set
sk, ck = tfhe.make_key_pair(rng, P, keygen="device") if a.params != "k2" else tfhe.make_key_pair(rng, P)
eng = ck.engine(0)
eng.set_option("pipeline_min", -1)
alts = [("default", [])] + [(kv, [(kv.split("=")[0], int(kv.split("=")[1]))]) for kv in a.ab]
for B in [int(v) for v in a.sizes.split(",")]:
    x = rng.integers(-2**31, 2**31, size=(B, P.lwe_size + 1), dtype=np.int64).astype(np.int32)
    ops = np.zeros(B, np.uint8)
    res, ref = {}, None
    for rep in range(a.reps + 1):
        for name, opts in alts:
            for k, v in opts: eng.set_option(k, v)
            out = eng.bootstrap(2**29, x, with_keyswitch=False)
            if rep: res.setdefault(name, []).append(eng.last_timing_ms(0))
            kern = eng.last_kernel_name()
            res.setdefault(name + "/kernel", kern)
            if ref is None: ref = out
            assert np.array_equal(out, ref), (B, name)
            for k, v in opts: eng.set_option(k, {"br_split": 1, "k2_rw": 0, "k2_w3": -1, "v3_rw": 0, "w2_rw": 0, "br_small": 1024, "br_prio_pct": 90}.get(k, 0))
    print(json.dumps({"params": a.params, "rotations": B, **{n: (round(float(np.median(v)), 3) if isinstance(v, list) else v) for n, v in res.items()}}), flush=True)
ck.close()
