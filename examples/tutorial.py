#!/usr/bin/env python3
"""The reference's examples/tutorial.jl on the MI355X engine: the encrypted minimum of 2017 and 42.

Same structure as the Julia example (prepare / process / verify, tutorial.jl:19-78); `process` builds the
comparator circuit once and runs it level by level on device-resident ciphertexts (tfhe_gates_level)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tfhe_jl_amd as tfhe


def int_to_bits(x, nbits=16):
    return [(x >> i) & 1 == 1 for i in range(nbits)]


def bits_to_int(bits):
    return sum(int(b) << i for i, b in enumerate(bits))


def prepare():
    rng = np.random.default_rng(123)
    secret_key, cloud_key = tfhe.make_key_pair(rng)                     # tutorial.jl:21-22
    ciphertext1 = tfhe.encrypt(rng, secret_key, int_to_bits(2017))      # :25-27
    ciphertext2 = tfhe.encrypt(rng, secret_key, int_to_bits(42))        # :30-32
    return secret_key, cloud_key, ciphertext1, ciphertext2


def encrypted_minimum_circuit(nb_bits=16):
    c = tfhe.Circuit()
    a, b = c.inputs(nb_bits), c.inputs(nb_bits)
    tmps1 = c.constant(False)                                           # :52
    for i in range(nb_bits):                                            # :54-56 with encrypted_compare_bit :42-45
        tmps1 = c.mux(c.xnor(a[i], b[i]), tmps1, a[i])
    c.set_outputs([c.mux(tmps1, b[i], a[i]) for i in range(nb_bits)])   # :60
    return c


def process(cloud_key, a, b):
    circuit = encrypted_minimum_circuit(len(a))
    inputs = tfhe.LweSampleArray(np.concatenate([a.data, b.data]))
    return circuit.run(cloud_key, inputs)


def verify(secret_key, answer):
    print(f"Answer: {bits_to_int(tfhe.decrypt(secret_key, answer))}")   # :72-77


if __name__ == "__main__":
    secret_key, cloud_key, ciphertext1, ciphertext2 = prepare()
    t0 = time.perf_counter()
    answer = process(cloud_key, ciphertext1, ciphertext2)
    print(f"(48 gates, 80 blind rotations, 18 levels: {1e3 * (time.perf_counter() - t0):.1f} ms incl. key upload)")
    verify(secret_key, answer)
