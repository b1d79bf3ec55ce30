#!/usr/bin/env python3
"""Runs ONE of the BASELINE.json configurations other than the headline bench line on ONE GPU and prints one JSON
object (kernel-side HIP-event times, roofline fraction of the blind-rotate kernel, decrypt check, rounding margin
and in-kernel clock from the DIAG instantiation of the same kernel).

  python tools/run_config.py --config 1|2host|3|4a|4b|5|k2 [--reps R] [--gates B] [--no-diag]

Also the program profiled for configs 4a / 4b / 5 (tools/profile.sh <tag> tools/run_config.py --config 4a)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe   # noqa: E402

HBM_PEAK = 8.0e12


def br_bytes(p, parties=1):
    if parties > 1:     # MK: P*n steps, per step (2 l P + 2 l) polys of N words   (98 304 000 B at P = 2)
        return parties * p.lwe_size * (2 * p.bs_decomp_length * parties + 2 * p.bs_decomp_length) * p.tlwe_polynomial_degree * 4
    return p.lwe_size * p.bs_decomp_length * (p.tlwe_mask_size + 1) ** 2 * p.tlwe_polynomial_degree * 4


OPTIONS = []


def apply_options(eng):
    for name, value in OPTIONS:
        eng.set_option(name, value)


def timed_calls(fn, reps):
    fn()
    wall = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); wall.append(time.perf_counter() - t0)
    return float(np.median(wall))


def diag(eng, fn, seconds=1.0):
    eng.set_option("measure_margin", 1)
    t0 = time.perf_counter()
    fn()
    while time.perf_counter() - t0 < seconds:
        fn()
    r = {"rounding_margin": eng.last_rounding_margin(), "kernel_clock_mhz": eng.last_kernel_clock_mhz()}
    eng.set_option("measure_margin", 0)
    return r


def single_key(params, B, reps, seed, want_diag, label):
    rng = np.random.default_rng(seed)
    sk, ck = tfhe.make_key_pair(rng, params)
    eng = ck.engine(0)
    apply_options(eng)
    bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.encrypt(rng, sk, bx).data, tfhe.encrypt(rng, sk, by).data
    ops = np.zeros(B, np.uint8)
    out = eng.gates(ops, x, y)
    br, ks = [], []
    def call():
        eng.gates(ops, x, y); br.append(eng.last_timing_ms(0)); ks.append(eng.last_timing_ms(1))
    wall = timed_calls(call, reps)
    brm = float(np.median(br[1:]))
    res = {"config": label, "options": dict(OPTIONS), "gates": B, "kernel": eng.last_kernel_name(), "blind_rotate_ms": brm, "keyswitch_ms": float(np.median(ks[1:])),
           "host_wall_ms": wall * 1e3, "gates_per_s_host_buffers": B / wall, "rot_per_s": B / brm * 1e3,
           "bytes_per_rotation": br_bytes(params), "frac_hbm_algorithmic": B / brm * 1e3 * br_bytes(params) / HBM_PEAK,
           "decrypt_ok_fraction": float((tfhe.decrypt(sk, out) == ~(bx & by)).mean())}
    if want_diag:
        res.update(diag(eng, lambda: eng.gates(ops, x, y)))
    ck.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", required=True, choices=["1", "2host", "3", "4a", "4b", "5", "k2", "mk4", "mk8"])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--gates", type=int, default=0)
    ap.add_argument("--no-diag", action="store_true")
    ap.add_argument("--lwe-size", type=int, default=0, help="experiment: override n (a shorter key that stays in L2 separates memory stalls from the rest)")
    ap.add_argument("--set", action="append", default=[], metavar="NAME=VALUE", help="engine option (tfhe_set_option) applied before the runs, e.g. --set n2048_rw=1")
    a = ap.parse_args()
    global OPTIONS
    OPTIONS = [(kv.split("=")[0], int(kv.split("=")[1])) for kv in a.set]
    if not any(k == "pipeline_min" for k, _ in OPTIONS):
        # this tool times KERNELS: one launch per batch (the two-stream split of large host-buffer batches, which hides PCIe
        # copies behind the other half's kernels, is what bench.py's value_pcie_inclusive measures; --set pipeline_min=4096 here)
        OPTIONS.append(("pipeline_min", -1))
    want_diag = not a.no_diag
    if a.config == "1":
        res = single_key(tfhe.tfhe_parameters_80(), 1, max(a.reps, 20), 123, want_diag, "1: single gate_nand, tfhe_parameters_80")
    elif a.config == "2host":
        res = single_key(tfhe.tfhe_parameters_80(), a.gates or 4096, a.reps, 123, want_diag, "2: 4096 NAND through host buffers")
    elif a.config == "4a":
        res = single_key(tfhe.tfhe_parameters_128(), a.gates or 4096, a.reps, 123, want_diag, "4a: tfhe_parameters_128, 4096 NAND")
    elif a.config == "4b":
        p = tfhe.SchemeParameters(a.lwe_size or 630, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)
        res = single_key(p, a.gates or 4096, a.reps, 2048, want_diag, "4b: synthetic N=2048 (n=630, l=3, beta=7), 4096 NAND")
    elif a.config == "k2":
        res = single_key(tfhe.tfhe_parameters_80(tlwe_mask_size=2), a.gates or 4096, a.reps, 77, want_diag, "tfhe_parameters_80(tlwe_mask_size=2), 4096 NAND")
    elif a.config == "3":
        rng = np.random.default_rng(123)
        sk, ck = tfhe.make_key_pair(rng, tfhe.tfhe_parameters_80())
        eng = ck.engine(0)
        for name, value in OPTIONS:
            if name != "pipeline_min" or ("pipeline_min=%d" % value) in a.set:   # (the tool's own pipeline_min=-1 default does not apply to this configuration)
                eng.set_option(name, value)
        B = a.gates or 8192              # one GPU's shard of 65536 / 8
        names = ["NAND", "AND", "OR", "XOR", "MUX"]
        mrng = np.random.default_rng(789)
        ops = np.array([tfhe.OPCODES[names[i]] for i in mrng.integers(0, 5, B)], np.uint8)
        ins = [tfhe.encrypt(rng, sk, mrng.integers(0, 2, B).astype(bool)).data for _ in range(3)]
        wall = timed_calls(lambda: eng.gates(ops, *ins), a.reps)      # as a caller gets it: two half-batches on two streams inside the call
        rotations = eng.last_rotation_count()
        eng.set_option("pipeline_min", -1)                            # kernel times: one launch of each kernel over the whole shard
        eng.gates(ops, *ins); eng.gates(ops, *ins)
        res = {"config": "3: mixed stream, one GPU's shard", "gates": B, "kernel": eng.last_kernel_name(), "gates_per_s_host_buffers": B / wall,
               "rotations": rotations, "blind_rotate_ms": eng.last_timing_ms(0), "keyswitch_ms": eng.last_timing_ms(1),
               "rot_per_s": rotations / eng.last_timing_ms(0) * 1e3, "host_wall_ms": wall * 1e3}
        ck.close()
    elif a.config in ("mk4", "mk8"):   # full-size 4- / 8-party sets (mk_api.jl:16-34), key expanded on the device
        p = tfhe.mktfhe_parameters_4party if a.config == "mk4" else tfhe.mktfhe_parameters_8party
        if a.lwe_size:
            p = tfhe.SchemeParameters(a.lwe_size, p.lwe_noise_stddev, 1024, 1, p.bs_decomp_length, p.bs_log2_base, p.bs_noise_stddev, 8, 2, p.ks_noise_stddev, p.max_parties)
        P = p.max_parties
        mrng = np.random.default_rng(321)
        t0 = time.perf_counter()
        sks = [tfhe.SecretKey(mrng, p) for _ in range(P)]
        shared = tfhe.SharedKey(mrng, p)
        parts = [tfhe.CloudKeyPart(mrng, s, shared) for s in sks]
        t_parts = time.perf_counter() - t0
        mck = tfhe.MKCloudKey(parts, expand="device")
        t0 = time.perf_counter()
        em = mck.engine(0)
        t_expand = time.perf_counter() - t0
        apply_options(em)
        B = a.gates or 1024
        m1, m2 = mrng.integers(0, 2, B).astype(bool), mrng.integers(0, 2, B).astype(bool)
        xm, ym = tfhe.mk_encrypt(mrng, sks, m1), tfhe.mk_encrypt(mrng, sks, m2)
        out = em.mk_gate_nand(xm, ym)
        br, ks = [], []
        def call():
            em.mk_gate_nand(xm, ym); br.append(em.last_timing_ms(0)); ks.append(em.last_timing_ms(1))
        wall = timed_calls(call, max(1, a.reps - 1))
        brm = float(np.median(br))
        res = {"config": f"{P}-party MK NAND (mk_api.jl:16-34), key expanded on the device", "gates": B, "kernel": em.last_kernel_name(),
               "host_keygen_parts_s": t_parts, "device_expand_and_load_s": t_expand, "blind_rotate_ms": brm, "keyswitch_ms": float(np.median(ks)),
               "host_wall_ms": wall * 1e3, "gates_per_s_host_buffers": B / wall, "rot_per_s": B / brm * 1e3, "bytes_per_rotation": br_bytes(p, P),
               "frac_hbm_algorithmic": B / brm * 1e3 * br_bytes(p, P) / HBM_PEAK,
               "decrypt_ok_fraction": float((tfhe.mk_decrypt(sks, out) == ~(m1 & m2)).mean())}
        if want_diag:
            res.update(diag(em, lambda: em.mk_gate_nand(xm[:64], ym[:64]), seconds=0.1))
        mck.close()
    else:   # 5: 2-party MK NAND
        p = tfhe.mktfhe_parameters_2party
        if a.lwe_size:
            p = tfhe.SchemeParameters(a.lwe_size, p.lwe_noise_stddev, 1024, 1, 4, 7, p.bs_noise_stddev, 8, 2, p.ks_noise_stddev, 2)
        mrng = np.random.default_rng(321)
        sks = [tfhe.SecretKey(mrng, p) for _ in range(2)]
        shared = tfhe.SharedKey(mrng, p)
        mck = tfhe.MKCloudKey([tfhe.CloudKeyPart(mrng, s, shared) for s in sks])
        B = a.gates or 1024
        m1, m2 = mrng.integers(0, 2, B).astype(bool), mrng.integers(0, 2, B).astype(bool)
        xm, ym = tfhe.mk_encrypt(mrng, sks, m1), tfhe.mk_encrypt(mrng, sks, m2)
        em = mck.engine(0)
        apply_options(em)
        out = em.mk_gate_nand(xm, ym)
        br, ks = [], []
        def call():
            em.mk_gate_nand(xm, ym); br.append(em.last_timing_ms(0)); ks.append(em.last_timing_ms(1))
        wall = timed_calls(call, a.reps)
        brm = float(np.median(br[1:]))
        res = {"config": "5: 2-party MK NAND", "options": dict(OPTIONS), "gates": B, "kernel": em.last_kernel_name(), "blind_rotate_ms": brm, "keyswitch_ms": float(np.median(ks[1:])),
               "host_wall_ms": wall * 1e3, "gates_per_s_host_buffers": B / wall, "rot_per_s": B / brm * 1e3, "bytes_per_rotation": br_bytes(p, 2),
               "frac_hbm_algorithmic": B / brm * 1e3 * br_bytes(p, 2) / HBM_PEAK,
               "decrypt_ok_fraction": float((tfhe.mk_decrypt(sks, out) == ~(m1 & m2)).mean())}
        if want_diag:
            res.update(diag(em, lambda: em.mk_gate_nand(xm, ym)))
        mck.close()
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
