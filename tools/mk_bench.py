import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import tfhe_jl_amd as tfhe
p = tfhe.mktfhe_parameters_2party
mrng = np.random.default_rng(321)
sks = [tfhe.SecretKey(mrng, p) for _ in range(2)]
shared = tfhe.SharedKey(mrng, p)
mck = tfhe.MKCloudKey([tfhe.CloudKeyPart(mrng, s, shared) for s in sks])
em = mck.engine(0)
for Bm in (1024, 4096):
    m1, m2 = mrng.integers(0, 2, Bm).astype(bool), mrng.integers(0, 2, Bm).astype(bool)
    xm, ym = tfhe.mk_encrypt(mrng, sks, m1), tfhe.mk_encrypt(mrng, sks, m2)
    out = em.mk_gate_nand(xm, ym)
    br = []
    for _ in range(3):
        em.mk_gate_nand(xm, ym); br.append(em.last_timing_ms(0))
    print(f"MK B={Bm}: blind rotate {np.median(br):.2f} ms  ({Bm/np.median(br)*1e3:.0f} rot/s), decrypt ok {float((tfhe.mk_decrypt(sks, out) == ~(m1 & m2)).mean()):.4f}", flush=True)
