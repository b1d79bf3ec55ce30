#!/usr/bin/env python3
"""The reference's examples/multikey.jl on the MI355X engine: a multi-key NAND.

    python examples/multikey.py [parties = 2 | 4 | 8]

The cloud key is expanded (RGSW.Expand, mk_internals.jl:304-345) on the GPU: 0.1-0.25 s even for the 2.4 GB 8-party key."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe

parties = int(sys.argv[1]) if len(sys.argv) > 1 else 2
params = {2: tfhe.mktfhe_parameters_2party, 4: tfhe.mktfhe_parameters_4party, 8: tfhe.mktfhe_parameters_8party}[parties]
rng = np.random.default_rng()

secret_keys = [tfhe.SecretKey(rng, params) for _ in range(parties)]            # on the clients (multikey.jl:13)
shared_key = tfhe.SharedKey(rng, params)                                       # created by the server (:16)
ck_parts = [tfhe.CloudKeyPart(rng, sk, shared_key) for sk in secret_keys]      # on the clients (:19)
cloud_key = tfhe.MKCloudKey(ck_parts, expand="device")                         # on the server (:23), expanded on its GPU

for trial in range(10):                                                        # :25
    mess1, mess2 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    enc1, enc2 = tfhe.mk_encrypt(rng, secret_keys, mess1), tfhe.mk_encrypt(rng, secret_keys, mess2)
    enc_out = tfhe.mk_gate_nand(cloud_key, enc1, enc2)                         # one GPU call
    dec_out = tfhe.mk_decrypt(secret_keys, enc_out)
    print(f"Trial {trial + 1}: {mess1} NAND {mess2} = {dec_out}  ({'ok' if dec_out == (not (mess1 and mess2)) else 'noise failure'})")
