#!/usr/bin/env python3
"""Per-phase cycle breakdown of the multi-wave blind-rotate kernels (development aid).
  make -C tfhe.jl_amd/csrc stamp && TFHE_MI355X_LIB=tfhe.jl_amd/lib/libtfhe_mi355x_stamp.so python tools/phase_profile.py --config 1|4b|5 [--gates B]
Prints, per wave of workgroup 0, the shader-clock ticks spent between consecutive STAMP marks, per step."""
import argparse, ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe

ap = argparse.ArgumentParser()
ap.add_argument("--config", required=True, choices=["1", "4b", "5"])
ap.add_argument("--gates", type=int, default=0)
ap.add_argument("--set", action="append", default=[])
a = ap.parse_args()
lib = tfhe._lib.load()
assert hasattr(lib, "tfhe_debug_phases"), "load the stamp build: TFHE_MI355X_LIB=.../libtfhe_mi355x_stamp.so"
lib.tfhe_debug_phases.argtypes = [C.c_void_p, C.c_void_p]

def phases(eng, steps, names):
    buf = np.zeros(64, np.uint64)
    rc = lib.tfhe_debug_phases(eng._h, buf.ctypes.data_as(C.c_void_p))
    assert rc == 0
    buf = buf.reshape(4, 16).astype(np.float64) / steps
    for w in range(4):
        if buf[w].sum() == 0: continue
        print(f"wave {w}: total {buf[w].sum():8.0f} cycles/step  " + "  ".join(f"{names[k] if k < len(names) else k}={buf[w][k]:.0f}" for k in range(16) if buf[w][k] > 0))

rng = np.random.default_rng(1)
if a.config == "1":
    sk, ck = tfhe.make_key_pair(rng, tfhe.tfhe_parameters_80())
    eng = ck.engine(0)
    B = a.gates or 1
    x, y = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data, tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
    for kv in a.set: eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    eng.set_option("measure_margin", 1)
    for _ in range(3): eng.gates(np.zeros(B, np.uint8), x, y)
    print(eng.last_kernel_name(), "BR ms", eng.last_timing_ms(0), "clock", eng.last_kernel_clock_mhz())
    if "h2" in eng.last_kernel_name():
        phases(eng, 500, ["rotate", "digits+combine", "fwdFFT256", "MAC+handoff-write", "barrierA", "read+add", "invFFT256", "swap-write+barrierR", "recombine+acc", "barrierE"])
    else:
        phases(eng, 500, ["rotate", "digits+fwdFFT", "keywait+MAC", "handoff-write", "barrier", "read+add", "invFFT", "untwist+acc"])
elif a.config == "4b":
    p = tfhe.SchemeParameters(630, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)
    sk, ck = tfhe.make_key_pair(rng, p)
    eng = ck.engine(0)
    B = a.gates or 4096
    x, y = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data, tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
    for kv in a.set: eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    eng.set_option("measure_margin", 1)
    for _ in range(2): eng.gates(np.zeros(B, np.uint8), x, y)
    print(eng.last_kernel_name(), "BR ms", eng.last_timing_ms(0), "clock", eng.last_kernel_clock_mhz())
    phases(eng, 630, ["rotate", "digits(x6)", "fwdFFT(x6)", "key+MAC(x6)", "(loop end)", "invFFT(x2)", "handoff-write+barrier", "handoff-read", "recombine", "end-barrier", "park+barrier+fetch (n2048x)"])
else:
    p = tfhe.mktfhe_parameters_2party
    sks = [tfhe.SecretKey(rng, p) for _ in range(2)]
    shared = tfhe.SharedKey(rng, p)
    mck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, s, shared) for s in sks])
    B = a.gates or 1024
    xm, ym = tfhe.mk_encrypt(rng, sks, rng.integers(0, 2, B).astype(bool)), tfhe.mk_encrypt(rng, sks, rng.integers(0, 2, B).astype(bool))
    em = mck.engine(0)
    for kv in a.set: em.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    em.set_option("measure_margin", 1)
    for _ in range(2): em.mk_gate_nand(xm, ym)
    print(em.last_kernel_name(), "BR ms", em.last_timing_ms(0), "clock", em.last_kernel_clock_mhz())
    phases(em, 500, ["rotate(x3)", "digits+fwdFFT(x6)", "key+MAC(x6)", "handoff-write", "barrierA", "read+add", "barrierB", "inv+untwist(x2|x1)", "barrierC"])
    print("(second party's 500 steps only; wave 0 owns the two masks, wave 1 the body)")
