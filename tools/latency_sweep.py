#!/usr/bin/env python3
"""Latency / small-batch sweep: host wall time and kernel times of B NAND gates for B = 1 .. 4096 with the
two-waves-per-rotation kernel enabled (default) and disabled (br_small = -1)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe
rng = np.random.default_rng(123)
sk, ck = tfhe.make_key_pair(rng, tfhe.tfhe_parameters_80())
eng = ck.engine(0)
res = []
for B in (1, 16, 64, 256, 512, 1024, 2048, 4096):
    x = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
    y = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
    ops = np.zeros(B, np.uint8)
    row = {"B": B}
    ref = None
    for name, small in (("w1", -1), ("w2", 1 << 30)):
        eng.set_option("br_small", small)
        out = eng.gates(ops, x, y)
        if ref is None: ref = out
        assert np.array_equal(out, ref)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); eng.gates(ops, x, y); t.append(time.perf_counter() - t0)
        row[name] = {"wall_ms": float(np.median(t)) * 1e3, "br_ms": eng.last_timing_ms(0), "ks_ms": eng.last_timing_ms(1)}
    res.append(row)
    print(json.dumps(row), flush=True)
