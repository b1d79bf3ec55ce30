#!/usr/bin/env python3
"""Times the BASELINE.json configurations other than the headline one on ONE GPU (kernel-side HIP events
plus host wall time, host buffers included): config 1 single NAND latency, config 3 mixed gate stream
(single-GPU shard), config 4a tfhe_parameters_128, config 5 two-party MK NAND.  Prints one JSON object."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe

def timed(fn, reps):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return float(np.median(t))

res = {}
rng = np.random.default_rng(123)
sk, ck = tfhe.make_key_pair(rng, tfhe.tfhe_parameters_80())
eng = ck.engine(0)
# config 1: single gate latency (B = 1 through the batch API)
x1, y1 = tfhe.encrypt(rng, sk, [True]).data, tfhe.encrypt(rng, sk, [False]).data
ops1 = np.zeros(1, np.uint8)
w = timed(lambda: eng.gates(ops1, x1, y1), 20)
res["config1_single_nand"] = {"host_wall_ms": w * 1e3, "blind_rotate_ms": eng.last_timing_ms(0), "keyswitch_ms": eng.last_timing_ms(1)}
# config 2 with host buffers (PCIe-inclusive)
B = 4096
x, y = tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data, tfhe.encrypt(rng, sk, rng.integers(0, 2, B).astype(bool)).data
ops = np.zeros(B, np.uint8)
w = timed(lambda: eng.gates(ops, x, y), 5)
res["config2_host_buffers"] = {"gates_per_s_pcie_inclusive": B / w, "host_wall_ms": w * 1e3, "device_ms": eng.last_timing_ms(2)}
# config 3: mixed stream, one GPU's shard of 65536/8 = 8192 gates
B3 = 8192
names = ["NAND", "AND", "OR", "XOR", "MUX"]
mrng = np.random.default_rng(789)
ops3 = np.array([tfhe.OPCODES[names[i]] for i in mrng.integers(0, 5, B3)], np.uint8)
ins3 = [tfhe.encrypt(rng, sk, mrng.integers(0, 2, B3).astype(bool)).data for _ in range(3)]
w = timed(lambda: eng.gates(ops3, *ins3), 3)
res["config3_mixed_8192_shard"] = {"gates_per_s": B3 / w, "rotations": eng.last_rotation_count(), "rotations_per_s_kernel": eng.last_rotation_count() / eng.last_timing_ms(0) * 1e3,
                                   "blind_rotate_ms": eng.last_timing_ms(0), "keyswitch_ms": eng.last_timing_ms(1), "host_wall_ms": w * 1e3}
ck.close()
# config 4a: 128-bit set
sk2, ck2 = tfhe.make_key_pair(rng, tfhe.tfhe_parameters_128())
e2 = ck2.engine(0)
x, y = tfhe.encrypt(rng, sk2, rng.integers(0, 2, B).astype(bool)).data, tfhe.encrypt(rng, sk2, rng.integers(0, 2, B).astype(bool)).data
w = timed(lambda: e2.gates(ops, x, y), 3)
br = e2.last_timing_ms(0)
res["config4a_128bit_4096"] = {"gates_per_s": B / w, "blind_rotate_ms": br, "keyswitch_ms": e2.last_timing_ms(1), "rot_per_s": B / br * 1e3,
                               "frac_hbm_algorithmic": B / br * 1e3 * 30965760 / 8e12}
ck2.close()
# config 4b: synthetic N = 2048 (630, l = 3, beta = 7)
p4b = tfhe.SchemeParameters(630, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)
sk3, ck3 = tfhe.make_key_pair(rng, p4b)
e3 = ck3.engine(0)
bx3, by3 = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
x, y = tfhe.encrypt(rng, sk3, bx3).data, tfhe.encrypt(rng, sk3, by3).data
w = timed(lambda: e3.gates(ops, x, y), 3)
out3 = e3.gates(ops, x, y)
br = e3.last_timing_ms(0)
res["config4b_synthetic_n2048_4096"] = {"gates_per_s": B / w, "blind_rotate_ms": br, "keyswitch_ms": e3.last_timing_ms(1), "rot_per_s": B / br * 1e3,
                                        "frac_hbm_algorithmic": B / br * 1e3 * 61931520 / 8e12,
                                        "decrypt_ok_fraction": float((tfhe.decrypt(sk3, out3) == ~(bx3 & by3)).mean())}
ck3.close()
# config 5: 2-party MK NAND x 1024
p = tfhe.mktfhe_parameters_2party
mrng = np.random.default_rng(321)
sks = [tfhe.SecretKey(mrng, p) for _ in range(2)]
shared = tfhe.SharedKey(mrng, p)
mck = tfhe.MKCloudKey([tfhe.CloudKeyPart(mrng, s, shared) for s in sks])
Bm = 1024
m1, m2 = mrng.integers(0, 2, Bm).astype(bool), mrng.integers(0, 2, Bm).astype(bool)
xm, ym = tfhe.mk_encrypt(mrng, sks, m1), tfhe.mk_encrypt(mrng, sks, m2)
em = mck.engine(0)
w = timed(lambda: em.mk_gate_nand(xm, ym), 3)
out = em.mk_gate_nand(xm, ym)
br = em.last_timing_ms(0)
res["config5_mk2_1024"] = {"gates_per_s": Bm / w, "blind_rotate_ms": br, "keyswitch_ms": em.last_timing_ms(1),
                           "decrypt_ok_fraction": float((tfhe.mk_decrypt(sks, out) == ~(m1 & m2)).mean()),
                           "frac_hbm_algorithmic": Bm / br * 1e3 * 98304000 / 8e12}
print(json.dumps(res, indent=1))
