#!/usr/bin/env python3
"""Batch-size sweep of the DEFAULT dispatch (no options): which blind-rotate kernel the engine picks for B NAND gates and
what it takes (kernel time by HIP events, host wall time through host buffers), 80-bit or 128-bit set.
  python tools/dispatch_sweep.py [--params 80|128]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe
ap = argparse.ArgumentParser(); ap.add_argument("--params", default="80"); a = ap.parse_args()
rng = np.random.default_rng(123)
sk, ck = tfhe.make_key_pair(rng, tfhe.tfhe_parameters_80() if a.params == "80" else tfhe.tfhe_parameters_128())
eng = ck.engine(0)
eng.set_option("pipeline_min", -1)        # single launches: kernel times are of the whole batch
print(f"| rotations | kernel | blind rotate ms | µs per rotation | host wall ms |\n|---|---|---|---|---|")
for B in (1, 16, 64, 256, 300, 512, 700, 1024, 1100, 1536, 2048, 2560, 3072, 4096, 5000, 8192, 9900, 16384):
    x = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
    y = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
    ops = np.zeros(B, np.uint8)
    eng.gates(ops, x, y)
    t, br = [], []
    for _ in range(7):
        t0 = time.perf_counter(); eng.gates(ops, x, y); t.append(time.perf_counter() - t0); br.append(eng.last_timing_ms(0))
    b = float(np.median(br))
    print(f"| {B} | `{eng.last_kernel_name().replace('blind_rotate_kernel_', '')}` | {b:.3f} | {1e3 * b / B:.2f} | {float(np.median(t)) * 1e3:.3f} |", flush=True)
