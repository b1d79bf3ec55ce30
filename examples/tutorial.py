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


def encrypted_minimum_circuit(nb_bits=16, log_depth=False):
    """log_depth=False: the reference's circuit gate for gate (a 16-deep XNOR -> MUX ripple, then 16 parallel MUXes:
    48 gates, 80 blind rotations, 18 levels).  log_depth=True: the same function with the ripple replaced by a reduction
    tree — more gates (63, 94 blind rotations) in 7 levels, for an engine whose cost is per LEVEL, not per gate."""
    c = tfhe.Circuit()
    a, b = c.inputs(nb_bits), c.inputs(nb_bits)
    if not log_depth:
        tmps1 = c.constant(False)                                       # :52
        for i in range(nb_bits):                                        # :54-56 with encrypted_compare_bit :42-45
            tmps1 = c.mux(c.xnor(a[i], b[i]), tmps1, a[i])
    else:
        # One ripple step is s -> MUX(t_i, s, a_i) with t_i = XNOR(a_i, b_i): "keep s if the bits agree, else take a_i".
        # Such maps compose: (later o earlier) keeps s iff both do (AND of the selectors) and otherwise yields
        # MUX(t_later, const_earlier, const_later).  Reduce the 16 steps pairwise, then apply the result to s = false:
        # MUX(T, false, A) = (not T) and A.
        steps = [(c.xnor(a[i], b[i]), a[i]) for i in range(nb_bits)]    # (selector, constant), least significant first
        while len(steps) > 1:
            nxt = [(c.and_(hi[0], lo[0]), c.mux(hi[0], lo[1], hi[1])) for lo, hi in zip(steps[0::2], steps[1::2])]
            if len(steps) % 2:
                nxt.append(steps[-1])
            steps = nxt
        tmps1 = c.andny(steps[0][0], steps[0][1])
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
    tree = encrypted_minimum_circuit(16, log_depth=True)
    inputs = tfhe.LweSampleArray(np.concatenate([ciphertext1.data, ciphertext2.data]))
    tree.run(cloud_key, inputs)
    t0 = time.perf_counter()
    answer = tree.run(cloud_key, inputs)
    print(f"(reduction tree: {len(tree.levels())} levels: {1e3 * (time.perf_counter() - t0):.1f} ms)")
    verify(secret_key, answer)
