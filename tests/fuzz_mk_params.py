#!/usr/bin/env python3
"""Multi-key parameter-space fuzz (as tests/fuzz_params.py; `python tests/fuzz_mk_params.py [cases] [seed]` on a GPU box): random
party counts (2 .. 9), N (16 .. 2048), l / beta, lwe_size, keyswitch length / base; mk_gate_nand on arbitrary words and on
encryptions against the oracle word for word, the DIAG instantiation's words and margin; every third case also expands the key
on the device (RGSW.Expand) and compares it with the host expansion."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tfhe_jl_amd as tfhe, oracle
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
t0 = time.time()
outside, worst = 0, (0.0, None)
for case in range(cases):
    parties = int(rng.choice([2, 2, 2, 3, 3, 4, 4, 5, 6, 8, 9]))
    N = int(2 ** rng.choice([4, 5, 6, 7, 8, 9, 10, 10, 10, 11]))
    beta = int(rng.integers(2, 9))
    l = int(rng.integers(1, min(32 // beta, 10) + 1))
    n = int(rng.integers(1, 7))
    gamma = int(rng.integers(1, 5))
    t = int(rng.integers(1, min(31 // gamma, 10) + 1))
    while n > 1 and parties * n * (2 * l * parties + 2 * l) * N > 3_000_000: n -= 1       # (host keygen time)
    p = tfhe.SchemeParameters(n, 0.012467, N, 1, l, beta, 3.29e-10, t, gamma, 2.44e-5, parties)
    sks = [tfhe.SecretKey(rng, p) for _ in range(parties)]
    shared = tfhe.SharedKey(rng, p)
    ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, sk, shared) for sk in sks], expand="host")
    o = oracle.Oracle(n, N, 1, l, beta, t, gamma, parties=parties)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    if case % 3 == 2:
        eng = tfhe.Engine(p, 0)
        expanded = eng.mk_expand_load_bootstrap_key(parties, *ck._part_arrays(), want_expanded=True)
        assert np.array_equal(expanded.reshape(ck.bootstrap_key.shape), ck.bootstrap_key), f"case {case}: device RGSW.Expand differs: {p}"
        eng.mk_load_keyswitch_key(ck.keyswitch_key, parties)
    else:
        eng = ck.engine(0)
    B = int(rng.choice([1, 2, 3, 5]))
    w = parties * n + 1
    x = rng.integers(-2**31, 2**31, size=(B, w), dtype=np.int64).astype(np.int32)
    y = rng.integers(-2**31, 2**31, size=(B, w), dtype=np.int64).astype(np.int32)
    x[0] = tfhe.mk_encrypt(rng, sks, [True])[0]
    y[0] = tfhe.mk_encrypt(rng, sks, [bool(rng.integers(0, 2))])[0]
    want = o.mk_gate_nand(x, y, nthreads=8)
    got = eng.mk_gate_nand(x, y)
    kern = eng.last_kernel_name()
    eng.set_option("measure_margin", 1)
    again = eng.mk_gate_nand(x, y)
    margin = eng.last_rounding_margin()
    eng.set_option("measure_margin", 0)
    # inside the Float64 exactness domain?  decided from the parameters alone (SchemeParameters.exactness), as tests/fuzz_params.py does
    cls, bound_log2, predicted = p.exactness()
    assert eng.get_option("exact_domain") == cls, (p, cls)
    if cls >= 1:
        assert margin < 0.25 and margin <= max(predicted, 0.02), f"case {case}: margin {margin} above the prediction {predicted} inside the exactness domain: {p} parties {parties} kernel {kern}"
    elif margin >= 0.25:
        outside += 1
        print(f"case {case:4d} P={parties} N={N:5d} l={l:2d} beta={beta} n={n} margin {margin:.4f} (predicted {predicted:.2f}, class 0): outside the Float64 domain, skipped  {kern}", flush=True)
        continue
    assert np.array_equal(got, want), f"case {case}: words differ (margin {margin}): {p} parties {parties} kernel {kern}"
    assert np.array_equal(again, want), f"case {case}: DIAG words differ: {p} parties {parties} kernel {kern}"
    if margin > worst[0]: worst = (margin, (parties, N, l, beta, n))
    print(f"case {case:4d} P={parties} N={N:5d} l={l:2d} beta={beta} n={n} t={t:2d} gamma={gamma} B={B} margin {margin:.4f}  {kern}" + ("  (key expanded on the device)" if case % 3 == 2 else ""), flush=True)
print(f"mk fuzz ok: {cases} parameter sets ({outside} outside the Float64 domain, skipped) in {time.time() - t0:.1f} s; largest rounding margin {worst[0]:.4f} at (parties, N, l, beta, n) = {worst[1]}")
