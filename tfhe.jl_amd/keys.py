"""Keys, encryption, decryption on the host — mirrors src/api.jl:84-169 with the constructors of
src/tlwe.jl:15-31, src/tgsw.jl:52-88, src/bootstrap.jl:1-16 and src/keyswitch.jl:7-42.

Key material is generated once per key pair with numpy (RNG-bound host work, as in the reference);
CloudKey keeps the flat arrays the C ABI takes (include/tfhe_mi355x.h) and uploads them to a
device context on first use.  The random stream is numpy's (PCG64), not Julia's MersenneTwister:
keys are data that crosses the boundary, the hot path itself consumes no randomness.
"""
import os

import numpy as np

from . import _lib
from .lwe import LweKey, LweSample, LweSampleArray, lwe_encrypt, lwe_encrypt_many, lwe_phase
from .numeric import (dtot32, encode_message, negacyclic_mul_binary, rand_gaussian_float,
                      rand_uniform_bool, rand_uniform_torus32, wrap32)
from .params import SchemeParameters, tfhe_parameters_80


class TLweKey:
    """tlwe.jl:11-21 — k binary polynomials."""

    def __init__(self, rng, N, k):
        self.N, self.k = N, k
        self.key = rand_uniform_bool(rng, k, N)

    def extract_lwe_key(self):
        """tlwe.jl:25-31"""
        return LweKey(None, self.k * self.N, key=self.key.reshape(-1))


def _tlwe_encrypt_zero_many(rng, alpha, tlwe_key: TLweKey, count):
    """`count` independent tlwe_encrypt_zero samples (tlwe.jl:63-73) -> int32 [count][k+1][N]."""
    N, k = tlwe_key.N, tlwe_key.k
    a_part = rand_uniform_torus32(rng, count, k, N)
    noise = dtot32(rng.standard_normal(size=(count, N)) * alpha).astype(np.int64)
    body = noise
    for c in range(k):
        body = body + negacyclic_mul_binary(tlwe_key.key[c], a_part[:, c, :]).astype(np.int64)
    return np.concatenate([a_part, wrap32(body)[:, None, :]], axis=1)


def make_bootstrap_key(rng, alpha, lwe_key: LweKey, tlwe_key: TLweKey, l, log2_base):
    """BootstrapKey (bootstrap.jl:6-15) in the canonical Int32 form [n][l][k+1][k+1][N]:
    key[i].samples[p, j].a[c], where samples = tgsw_encrypt(s_i) (tgsw.jl:84-88):
    l*(k+1) zero encryptions plus s_i * 2^(32 - p*beta) on the constant term of component j of row (p, j)
    (tgsw.jl:52-72; `Polynomial + scalar` adds to coefficient 0)."""
    n, N, k = lwe_key.size, tlwe_key.N, tlwe_key.k
    bk = _tlwe_encrypt_zero_many(rng, alpha, tlwe_key, n * l * (k + 1)).reshape(n, l, k + 1, k + 1, N)
    s = lwe_key.key.astype(np.int64)
    for p in range(l):
        gadget = np.int64(1) << (32 - (p + 1) * log2_base)
        for j in range(k + 1):
            bk[:, p, j, j, 0] = wrap32(bk[:, p, j, j, 0].astype(np.int64) + s * gadget)
    return np.ascontiguousarray(bk, np.int32)


def make_keyswitch_key(rng, alpha, t, log2_base, out_key: LweKey, tlwe_key: TLweKey):
    """KeyswitchKey (keyswitch.jl:14-41) as Int32 [kN][t][base-1][n+1] (= key[h, j, i] in Julia)."""
    in_key = tlwe_key.extract_lwe_key()
    kN, n = in_key.size, out_key.size
    base = 1 << log2_base
    noise = rand_gaussian_float(rng, alpha, kN, t, base - 1)
    noise -= noise.sum() / noise.size                                        # keyswitch.jl:29
    i = np.arange(kN)[:, None, None]
    j = np.arange(1, t + 1)[None, :, None]
    h = np.arange(1, base)[None, None, :]
    message = (in_key.key.astype(np.int64)[i] * h) << (32 - j * log2_base)   # keyswitch.jl:35
    a = rand_uniform_torus32(rng, kN, t, base - 1, n)
    dot = (a.reshape(-1, n).astype(np.int64) @ out_key.key.astype(np.int64)).reshape(kN, t, base - 1)
    b = wrap32(message + dtot32(noise).astype(np.int64) + dot)               # lwe.jl:49-55
    return np.ascontiguousarray(np.concatenate([a, b[..., None]], axis=-1), np.int32)


class SecretKey:
    """api.jl:92-100"""

    def __init__(self, rng, params: SchemeParameters):
        self.params = params
        self.key = LweKey(rng, params.lwe_size)
        self.cloud_keygen_seed = None     # set by CloudKey(keygen="device"): it regenerates the key's noise, so it lives HERE


class CloudKey:
    """api.jl:111-127.  Holds the flat key arrays; `engine(device)` gives the device context."""

    def __init__(self, rng, secret_key: SecretKey, keygen="host", device=0, noise_seed=None):
        """keygen="host": numpy (the reference does this work on the host too); keygen="device": the key material is
        generated on GPU `device` (tfhe_keygen_cloud_key) and the context that made it stays loaded as `engine(device)`.
        The TLWE key bits and the two seed words that key the PUBLIC mask streams come from `rng`; the four words (128
        bits) that key the NOISE streams come from the operating system's generator (os.urandom) whatever `rng` is —
        numpy's generators are statistical, not cryptographic, and Philox, which the library expands this secret with, makes
        no secrecy claim of its own.  `noise_seed` (four 32-bit words) overrides that for reproducible tests.  The seed is as
        secret as the secret key (it regenerates every noise term): it is kept on the SecretKey object, never on this
        (public) one."""
        p = secret_key.params
        self.params = p
        self._engines = {}
        tlwe_key = TLweKey(rng, p.tlwe_polynomial_degree, p.tlwe_mask_size)
        if keygen == "device":
            mask_words = rng.integers(0, 2**32, 2, dtype=np.uint64).astype(np.uint32)
            noise_words = (np.frombuffer(os.urandom(16), dtype=np.uint32) if noise_seed is None
                           else np.ascontiguousarray(noise_seed, dtype=np.uint32).reshape(-1))
            if noise_words.size != 4:
                raise ValueError("noise_seed must be four 32-bit words")
            seed = np.concatenate([mask_words, noise_words]).astype(np.uint32)
            e = _lib.Engine(p, device) if np.ndim(device) == 0 else _lib.Engine(p, devices=[int(d) for d in device])
            try:
                self.bootstrap_key, self.keyswitch_key = e.keygen_cloud_key(secret_key.key.key, tlwe_key.key, p.bs_noise_stddev,
                                                                            p.ks_noise_stddev, seed)
            except Exception:
                e.close()
                raise
            self.bootstrap_key = self.bootstrap_key.reshape(p.lwe_size, p.bs_decomp_length, p.tlwe_mask_size + 1,
                                                            p.tlwe_mask_size + 1, p.tlwe_polynomial_degree)
            self._engines[device if np.ndim(device) == 0 else tuple(int(d) for d in device)] = e
            secret_key.cloud_keygen_seed = seed
            return
        if keygen != "host":
            raise ValueError("keygen must be 'host' or 'device'")
        self.bootstrap_key = make_bootstrap_key(rng, p.bs_noise_stddev, secret_key.key, tlwe_key,
                                                p.bs_decomp_length, p.bs_log2_base)
        self.keyswitch_key = make_keyswitch_key(rng, p.ks_noise_stddev, p.ks_decomp_length, p.ks_log2_base,
                                                secret_key.key, tlwe_key)

    def engine(self, device=0) -> "_lib.Engine":
        """`device`: a device id, or a sequence of ids for a multi-device context (every batch call is then split
        over those GPUs inside the library: the analogue of `gate_nand.(ck, xs, ys)` over a whole node)."""
        key = device if np.ndim(device) == 0 else tuple(int(d) for d in device)
        e = self._engines.get(key)
        if e is None:
            e = _lib.Engine(self.params, device) if np.ndim(device) == 0 else _lib.Engine(self.params, devices=list(key))
            e.load_bootstrap_key(self.bootstrap_key)
            e.load_keyswitch_key(self.keyswitch_key)
            self._engines[key] = e
        return e

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}


def make_key_pair(rng, params: SchemeParameters = None, keygen="host", device=0, noise_seed=None):
    """api.jl:139-146 (keygen="device": the cloud key is generated on the GPU, see CloudKey)"""
    if params is None:
        params = tfhe_parameters_80()
    secret_key = SecretKey(rng, params)
    cloud_key = CloudKey(rng, secret_key, keygen=keygen, device=device, noise_seed=noise_seed)
    return secret_key, cloud_key


def encrypt(rng, key: SecretKey, message):
    """api.jl:155-158.  A bool gives an LweSample; an array of bools gives an LweSampleArray."""
    alpha = key.params.lwe_noise_stddev
    if np.ndim(message) == 0:
        return lwe_encrypt(rng, encode_message(1 if message else -1, 8), alpha, key.key)
    bits = np.asarray(message, bool).reshape(-1)
    mu = np.where(bits, encode_message(1, 8), encode_message(-1, 8))
    return LweSampleArray(lwe_encrypt_many(rng, mu, alpha, key.key))


def decrypt(key: SecretKey, sample):
    """api.jl:167-169 — phase > 0."""
    if isinstance(sample, LweSample):
        return bool(lwe_phase(sample.flat(), key.key) > 0)
    data = sample.data if isinstance(sample, LweSampleArray) else np.asarray(sample, np.int32)
    return lwe_phase(data, key.key) > 0
