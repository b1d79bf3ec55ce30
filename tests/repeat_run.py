#!/usr/bin/env python3
"""Repeat run (under tests/ because the oracle is its checker; not collected by pytest — `python tests/repeat_run.py [launches]` on a GPU
box): the two kernels whose waves hand blocks over through polled LDS words (blind_rotate_kernel_n2048x, mk_blind_rotate_kernel_w2:
pair_signal, kernels_common.hpp) at full size, the same inputs launched again and again: EVERY output word of every launch must equal the
first launch's (a hand-off read too early would show as a word that differs in one launch), and the first launch is compared with the
oracle on a sample of rows.  A launch of config 4b is 4096 rotations x 630 steps x 2 hand-offs = 5.2 M hand-offs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import tfhe_jl_amd as tfhe, oracle
from conftest import KeySet
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = time.time()

# config 4b's shape (N = 2048, n = 630, l = 3, beta = 7), batch sizes around the workgroup pairing
p = tfhe.SchemeParameters(630, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)
K = KeySet(tfhe, oracle, p, seed=2048)
eng = K.ck.engine(0)
rng = np.random.default_rng(7)
for B in (4096, 4095, 2049, 777):
    bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.encrypt(K.rng, K.sk, bx).data, tfhe.encrypt(K.rng, K.sk, by).data
    ops = np.zeros(B, np.uint8)
    first = eng.gates(ops, x, y)
    assert eng.last_kernel_name() == "blind_rotate_kernel_n2048x<3,rw2>", eng.last_kernel_name()
    rows = rng.choice(B, 48, replace=False)
    want = K.oracle.gates(ops[rows], x[rows], y[rows], None, nthreads=16)
    assert np.array_equal(first[rows], want), f"N = 2048, {B} gates: differs from the oracle"
    assert np.array_equal(tfhe.decrypt(K.sk, first), ~(bx & by))
    n = launches if B == 4096 else max(3, launches // 4)
    for i in range(n):
        again = eng.gates(ops, x, y)
        bad = np.flatnonzero((again != first).any(axis=1))
        assert bad.size == 0, f"N = 2048, {B} gates, launch {i + 1}: rows {bad[:8]} differ from the first launch"
    print(f"N = 2048: {B} gates x {n + 1} launches identical, 48 rows == oracle ({time.time() - t0:.0f} s)", flush=True)
K.ck.close()

# config 5 (mktfhe_parameters_2party)
p2 = tfhe.mktfhe_parameters_2party
mrng = np.random.default_rng(11)
sks = [tfhe.SecretKey(mrng, p2) for _ in range(2)]
shared = tfhe.SharedKey(mrng, p2)
ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(mrng, sk, shared) for sk in sks])
o = oracle.Oracle(p2.lwe_size, 1024, 1, p2.bs_decomp_length, p2.bs_log2_base, 8, 2, parties=2)
o.load_bootstrap_key(ck.bootstrap_key)
o.load_keyswitch_key(ck.keyswitch_key)
eng = ck.engine(0)
for B in (1024, 1023, 515):
    m1, m2 = mrng.integers(0, 2, B).astype(bool), mrng.integers(0, 2, B).astype(bool)
    x, y = tfhe.mk_encrypt(mrng, sks, m1), tfhe.mk_encrypt(mrng, sks, m2)
    first = eng.mk_gate_nand(x, y)
    assert eng.last_kernel_name() == "mk_blind_rotate_kernel_w2<4>", eng.last_kernel_name()
    rows = mrng.choice(B, 32, replace=False)
    assert np.array_equal(first[rows], o.mk_gate_nand(x[rows], y[rows], nthreads=16)), f"2 parties, {B} gates: differs from the oracle"
    n = launches if B == 1024 else max(3, launches // 4)
    for i in range(n):
        again = eng.mk_gate_nand(x, y)
        bad = np.flatnonzero((again != first).any(axis=1))
        assert bad.size == 0, f"2 parties, {B} gates, launch {i + 1}: rows {bad[:8]} differ from the first launch"
    print(f"2 parties: {B} gates x {n + 1} launches identical, 32 rows == oracle ({time.time() - t0:.0f} s)", flush=True)
ck.close()
print(f"repeat run ok: {time.time() - t0:.1f} s")
