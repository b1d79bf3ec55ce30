#!/usr/bin/env python3
"""Parameter-space fuzz (lives under tests/ because the oracle is the checker; not collected by pytest — run it as
`python tests/fuzz_params.py [cases] [seed]` on a GPU box): random parameter sets over everything tfhe_ctx_create accepts —
N any power of two 2 .. 4096 (8192 now and then), tlwe_mask_size 1 .. 6, any l / beta with l beta <= 32, lwe_size 1 .. 40,
any keyswitch length / base with t gamma <= 31 — each with a random batch of arbitrary input words (edge words included):
blind rotation + extraction against the oracle word for word, whole gates (random opcodes) against the oracle, the DIAG
instantiation's words and margin, and the kernel the dispatcher chose printed.  Rounding-margin violations are reported, not
asserted, for sets whose noiseless worst case exceeds the Float64 transform's range by construction (l beta = 32 at large N k)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tfhe_jl_amd as tfhe, oracle
from conftest import KeySet
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
names = list(tfhe.OPCODES)
MU = 2**29
t0 = time.time()
worst = (0.0, None)
outside = 0
for case in range(cases):
    logN = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 10, 10, 10, 11, 11, 12, 13], p=None))
    N = 2**logN
    k = int(rng.choice([1, 1, 1, 2, 2, 3, 4, 5, 6]))
    if N * k > 16384: k = max(1, 16384 // N)
    beta = int(rng.integers(1, 13))
    l = int(rng.integers(1, min(32 // beta, 8) + 1))
    n = int(rng.integers(1, 24 if N >= 2048 else 40))
    gamma = int(rng.integers(1, 6))
    t = int(rng.integers(1, min(31 // gamma, 12) + 1))
    while n > 1 and n * l * (k + 1) ** 2 * N > 6_000_000: n -= 1                         # (host keygen time, not an engine limit)
    while t > 1 and k * N * t * (2**gamma - 1) * (n + 1) > 12_000_000: t -= 1
    p = tfhe.SchemeParameters(n, 1 / 2**15, N, k, l, beta, 9e-9, t, gamma, 1 / 2**15, 1)
    K = KeySet(tfhe, oracle, p, seed=int(rng.integers(1, 2**31)))
    eng = K.ck.engine(0)
    B = int(rng.choice([1, 2, 3, 5, 9, 17]))
    x = rng.integers(-2**31, 2**31, size=(B, n + 1), dtype=np.int64).astype(np.int32)
    edge = np.array([2**31 - 1, -2**31, 2**20, 2**20 - 1, -2**20, -2**20 - 1, 0, 1, -1], np.int64).astype(np.int32)
    x[0, :] = edge[rng.integers(0, len(edge), n + 1)]
    want = K.oracle.bootstrap(MU, x, with_keyswitch=False, nthreads=8)
    got = eng.bootstrap(MU, x, with_keyswitch=False)
    kern = eng.last_kernel_name()
    eng.set_option("measure_margin", 1)
    again = eng.bootstrap(MU, x, with_keyswitch=False)
    margin = eng.last_rounding_margin()
    eng.set_option("measure_margin", 0)
    # Is the set inside the Float64 exactness domain?  Decided WITHOUT the engine under test: SchemeParameters.exactness() is the
    # a-priori law of (N, k, l, beta), which the engine must restate (tfhe_get_option "exact_domain").  Inside it
    # (class 1 or 2) nothing is skipped: the measured margin must stay below 1/4 and below the prediction, and every word is
    # compared.  Outside (class 0) the words are compared only where the measured margin still allows.
    cls, bound_log2, predicted = p.exactness()
    assert eng.get_option("exact_domain") == cls and abs(eng.get_option("exact_margin_x1e6") / 1e6 - predicted) < 1e-5 + 1e-6 * predicted, (p, cls, predicted)
    if cls >= 1:
        assert margin < 0.25 and margin <= max(predicted, 0.02), f"case {case}: margin {margin} above the prediction {predicted} inside the exactness domain: {p} kernel {kern}"
    elif margin >= 0.25:
        # a pre-rounding value this far from an integer: the set is outside what a Float64 transform computes exactly (the
        # reference's own FFTW transform rounds such values its own way, polynomials.jl:115-116) — nothing to compare
        outside += 1
        print(f"case {case:4d} N={N:5d} k={k} l={l:2d} beta={beta:2d} n={n:2d} margin {margin:.4f} (predicted {predicted:.2f}, class 0): outside the Float64 domain, skipped  {kern}", flush=True)
        K.ck.close()
        continue
    assert np.array_equal(got, want), f"case {case}: rotation words differ (margin {margin}): {p} kernel {kern}"
    assert np.array_equal(again, want), f"case {case}: DIAG words differ: {p} kernel {kern}"
    if margin > worst[0]: worst = (margin, (N, k, l, beta, n))
    sel = rng.integers(0, len(names), B)
    ops = np.array([tfhe.OPCODES[names[s_]] for s_ in sel], np.uint8)
    ins = [rng.integers(-2**31, 2**31, size=(B, n + 1), dtype=np.int64).astype(np.int32) for _ in range(3)]
    gw = K.oracle.gates(ops, *ins, nthreads=8)
    gg = eng.gates(ops, *ins)
    assert np.array_equal(gg, gw), f"case {case}: gate words differ: {p} kernel {kern} / {eng.last_kernel_name()}"
    # the oracle's own Float64 transform (radix-2, one rounding per output as the reference) measures a margin of the same size
    om = K.oracle.last_margin
    assert cls == 0 or om < 0.25, (case, om)
    # (printed beside the engine's margin, not compared with it: the two come from different inputs and, at small lwe_size, from a
    #  handful of CMUX steps — 0.0017 against 0.156 at lwe_size 2 in one run; what IS asserted is that neither leaves the domain the
    #  a-priori class promises)
    print(f"case {case:4d} N={N:5d} k={k} l={l:2d} beta={beta:2d} n={n:2d} t={t:2d} gamma={gamma} B={B:2d} margin {margin:.4f} (oracle {om:.4f}, predicted {predicted:.4f}, class {cls})  {kern}", flush=True)
    K.ck.close()
print(f"fuzz ok: {cases} parameter sets ({outside} outside the Float64 domain, skipped) in {time.time() - t0:.1f} s; largest rounding margin {worst[0]:.4f} at (N, k, l, beta, n) = {worst[1]}")
