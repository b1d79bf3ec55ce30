"""LWE samples on the host — mirrors src/lwe.jl (the parts the API boundary exposes)."""
import numpy as np

from .numeric import rand_gaussian_torus32, rand_uniform_bool, rand_uniform_torus32, wrap32


class LweKey:
    """lwe.jl:6-15"""

    def __init__(self, rng, size, key=None):
        self.size = size
        self.key = rand_uniform_bool(rng, size) if key is None else np.asarray(key, np.int32)


class LweSample:
    """An encrypted bit, lwe.jl:21-29.  Flat layout = a[0..n-1], b (include/tfhe_mi355x.h)."""
    __slots__ = ("a", "b", "current_variance")

    def __init__(self, a, b, current_variance=0.0):
        self.a = np.asarray(a, np.int32)
        self.b = np.int32(b)
        self.current_variance = float(current_variance)

    @property
    def size(self):
        return self.a.size

    def flat(self):
        return np.concatenate([self.a, np.array([self.b], np.int32)])

    @staticmethod
    def from_flat(w, current_variance=0.0):
        w = np.asarray(w, np.int32)
        return LweSample(w[:-1].copy(), w[-1], current_variance)


class LweSampleArray:
    """A vector of LWE samples stored as one int32 [B][n+1] matrix — what Julia code writes as
    `Vector{LweSample}` and broadcasts gates over (docs/src/manual.md:28-35)."""

    def __init__(self, data):
        self.data = np.ascontiguousarray(data, np.int32)
        assert self.data.ndim == 2

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, i):
        if isinstance(i, (int, np.integer)):
            return LweSample.from_flat(self.data[i])
        return LweSampleArray(self.data[i])

    @staticmethod
    def from_samples(samples):
        return LweSampleArray(np.stack([s.flat() for s in samples]))


def lwe_encrypt(rng, message, alpha, key: LweKey):
    """lwe.jl:38-43 — b = gaussian(message, alpha) + <a, s>"""
    a = rand_uniform_torus32(rng, key.size)
    b = wrap32(np.int64(rand_gaussian_torus32(rng, message, alpha)) + np.sum(a.astype(np.int64) * key.key))
    return LweSample(a, b, alpha**2)


def lwe_encrypt_many(rng, messages, alpha, key: LweKey):
    """Vectorised lwe_encrypt over an array of Torus32 messages -> int32 [B][n+1]."""
    messages = np.asarray(messages, np.int64)
    B = messages.size
    a = rand_uniform_torus32(rng, B, key.size)
    noise = rand_gaussian_torus32(rng, 0, alpha, B).astype(np.int64)
    b = wrap32(messages + noise + (a.astype(np.int64) @ key.key.astype(np.int64)))
    return np.concatenate([a, b[:, None]], axis=1).astype(np.int32)


def lwe_phase(flat, key: LweKey):
    """lwe.jl:59 — phase = b - <a, s>, for one sample or a [B][n+1] matrix."""
    flat = np.asarray(flat, np.int32)
    a, b = flat[..., :-1].astype(np.int64), flat[..., -1].astype(np.int64)
    return wrap32(b - a @ key.key.astype(np.int64))
