#!/usr/bin/env python3
"""Mints tests/golden/kat_n4.npz: a reduced-n key set, a handful of input LWE words and the outputs
of the ORACLE (oracle/tfhe_oracle.c, reference-style FFT back-end, cross-checked here against its exact
back-end).  RESTATEMENT-DERIVED: the Julia reference cannot run in the build image (no Julia, un-vendored
DarkIntegers/FFTW.jl) and ships no Int32 fixtures of its own; regenerate from the real reference and diff
if a Julia runtime ever becomes available.

    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import tfhe_jl_amd as tfhe  # noqa: E402

n, N, k, l, beta, t, g = 4, 1024, 1, 2, 10, 8, 2
params = tfhe.SchemeParameters(n, 1 / 2**15, N, k, l, beta, 9e-9, t, g, 1 / 2**15, 1)
rng = np.random.default_rng(20261003)
sk, ck = tfhe.make_key_pair(rng, params)
o = oracle.Oracle(n, N, k, l, beta, t, g)
o.load_bootstrap_key(ck.bootstrap_key)
o.load_keyswitch_key(ck.keyswitch_key)

names = ["NAND", "OR", "AND", "XOR", "XNOR", "NOT", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX", "CONST0", "CONST1", "COPY"]
ops = np.array([tfhe.OPCODES[x] for x in names] * 2, np.uint8)
B = ops.size
ins = [rng.integers(-2**31, 2**31, size=(B, n + 1), dtype=np.int64).astype(np.int32) for _ in range(3)]
ins[0][0, :] = [2**31 - 1, -2**31, 2**20, -2**20 - 1, 0]       # mod-switch edge words
out = o.gates(ops, *ins)
assert np.array_equal(out, o.gates(ops, *ins, mode=oracle.MODE_EXACT)), "FFT and exact back-ends disagree"
ext = o.bootstrap(2**29, ins[0][:8], with_keyswitch=False)
assert np.array_equal(ext, o.bootstrap(2**29, ins[0][:8], with_keyswitch=False, mode=oracle.MODE_EXACT))
ks_out = o.keyswitch(ext)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_n4.npz")
np.savez_compressed(path, params=np.array([n, N, k, l, beta, t, g], np.int32), secret_key=sk.key.key,
                    bootstrap_key=ck.bootstrap_key, keyswitch_key=ck.keyswitch_key, ops=ops,
                    in0=ins[0], in1=ins[1], in2=ins[2], out=out, ext=ext, ks_out=ks_out)
print("wrote", path, os.path.getsize(path), "bytes")
