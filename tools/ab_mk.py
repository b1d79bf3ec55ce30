#!/usr/bin/env python3
"""Interleaved A/B of an engine option on the multi-key NAND (one process, one device), words compared:
  python tools/ab_mk.py --parties 2|4|8 --ab NAME=VALUE [--gates 1024] [--reps 5]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe
ap = argparse.ArgumentParser()
ap.add_argument("--parties", type=int, default=2)
ap.add_argument("--ab", action="append", default=[])
ap.add_argument("--gates", type=int, default=1024)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
p = {2: tfhe.mktfhe_parameters_2party, 4: tfhe.mktfhe_parameters_4party, 8: tfhe.mktfhe_parameters_8party}[a.parties]
rng = np.random.default_rng(321)
sks = [tfhe.SecretKey(rng, p) for _ in range(a.parties)]
shared = tfhe.SharedKey(rng, p)
ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, s, shared) for s in sks], expand="device")
eng = ck.engine(0)
B = a.gates
x, y = tfhe.mk_encrypt(rng, sks, rng.integers(0, 2, B).astype(bool)), tfhe.mk_encrypt(rng, sks, rng.integers(0, 2, B).astype(bool))
alts = [("default", [])] + [(kv, [(kv.split("=")[0], int(kv.split("=")[1]))]) for kv in a.ab]
res, ref = {}, None
for rep in range(a.reps + 1):
    for name, opts in alts:
        for k, v in opts: eng.set_option(k, v)
        out = eng.mk_gate_nand(x, y)
        if rep: res.setdefault(name, []).append(eng.last_timing_ms(0))
        res[name + "/kernel"] = eng.last_kernel_name()
        if ref is None: ref = out
        assert np.array_equal(out, ref), name
        for k, v in opts: eng.set_option(k, 0)
print(json.dumps({"parties": a.parties, "gates": B, **{n: (round(float(np.median(v)), 3) if isinstance(v, list) else v) for n, v in res.items()}}))
