#!/usr/bin/env python3
"""Latency / batch-size sweep: host wall time and kernel times of B NAND gates for B = 1 .. 16384 with each of the
three blind-rotate kernels forced (v3: one wave per rotation; w2: two waves; h2: every transform over two waves)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe
rng = np.random.default_rng(123)
sk, ck = tfhe.make_key_pair(rng, tfhe.tfhe_parameters_80())
eng = ck.engine(0)
res = []
for B in (1, 8, 16, 64, 256, 512, 768, 1024, 1536, 2048, 3072, 4096, 8192, 16384):
    x = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
    y = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
    ops = np.zeros(B, np.uint8)
    row = {"B": B}
    ref = None
    for name, small, tiny in (("v3", -1, -1), ("w2", 1 << 30, -1), ("h2", 1 << 30, 1 << 30)):
        if name == "h2" and B > 64:
            continue
        eng.set_option("br_small", small)
        eng.set_option("br_tiny", tiny)
        out = eng.gates(ops, x, y)
        if ref is None: ref = out
        assert np.array_equal(out, ref)
        t = []
        for _ in range(5):
            t0 = time.perf_counter(); eng.gates(ops, x, y); t.append(time.perf_counter() - t0)
        row[name] = {"wall_ms": float(np.median(t)) * 1e3, "br_ms": eng.last_timing_ms(0), "ks_ms": eng.last_timing_ms(1)}
    res.append(row)
    print(json.dumps(row), flush=True)
