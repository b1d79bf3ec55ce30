"""Host-side torus numerics (numpy) — mirrors src/numeric-functions.jl and the polynomial product the
key generator needs (src/polynomials.jl:142-144).  Key generation and encrypt/decrypt stay on the
host (one-off, RNG-bound work); the bootstrapping hot path never comes through here."""
import numpy as np

Torus32 = np.int32


def rand_uniform_bool(rng, *dims):
    """numeric-functions.jl:4-6"""
    return rng.integers(0, 2, size=dims, dtype=np.int32)


def rand_uniform_torus32(rng, *dims):
    """numeric-functions.jl:9-11"""
    return rng.integers(-2**31, 2**31, size=dims, dtype=np.int64).astype(np.int32)


def rand_gaussian_float(rng, sigma, *dims):
    """numeric-functions.jl:14-16"""
    return rng.standard_normal(size=dims) * sigma


def dtot32(d):
    """numeric-functions.jl:51-53 — trunc(Int32, d * 2^32) for d in [-0.5, 0.5)."""
    return np.trunc(np.asarray(d, dtype=np.float64) * 2.0**32).astype(np.int64).astype(np.int32)


def rand_gaussian_torus32(rng, message, sigma, *dims):
    """numeric-functions.jl:20-23"""
    err = rng.standard_normal(size=dims) * sigma
    return wrap32(np.int64(message) + dtot32(err).astype(np.int64))


def wrap32(x):
    """Int64 -> Int32 two's-complement wrap (Julia Int32 arithmetic wraps)."""
    return (np.asarray(x, dtype=np.int64) & 0xFFFFFFFF).astype(np.uint32).astype(np.int32)


def encode_message(mu, message_space):
    """numeric-functions.jl:42-45"""
    log2_ms = int(message_space).bit_length() - 1
    return int(wrap32(np.int64(mu) << (32 - log2_ms)))


def decode_message(phase, message_space):
    """numeric-functions.jl:31-34"""
    log2_ms = int(message_space).bit_length() - 1
    p = wrap32(np.asarray(phase, dtype=np.int64) + (1 << (32 - log2_ms - 1)))
    return p >> (32 - log2_ms)


def negacyclic_mul_binary(s, a):
    """(s * a) mod (X^N + 1, 2^32) for a binary (0/1) polynomial s and Int32 polynomials a[..., N].

    Plays the role of transformed_mul (polynomials.jl:142-144) in key generation
    (tlwe.jl:69-71).  Exact: a is split into 16-bit halves so every FFT value stays below 2^27.
    """
    s = np.asarray(s, dtype=np.int64)
    a = np.asarray(a)
    N = s.shape[-1]
    au = a.astype(np.int64) & 0xFFFFFFFF
    lo, hi = au & 0xFFFF, au >> 16
    j = np.arange(N)
    tw = np.exp(1j * np.pi * j / N)
    fs = np.fft.fft(s * tw, axis=-1)

    def prod(x):
        y = np.fft.ifft(np.fft.fft(x * tw, axis=-1) * fs, axis=-1) * np.conj(tw)
        return np.rint(y.real).astype(np.int64)

    return wrap32(prod(lo) + (prod(hi) << 16))


def negacyclic_mul_small(d, a):
    """(d * a) mod (X^N + 1, 2^32) for small-integer polynomials d[..., N] (|d| <= 2^12, e.g. gadget
    digits or binary keys) and Int32 polynomials a[..., N], broadcasting over leading axes.
    Exact: a is split into 16-bit halves so every FFT value stays below 2^40 (float64 has 53 bits)."""
    d = np.asarray(d, dtype=np.float64)
    a = np.asarray(a)
    N = d.shape[-1]
    au = a.astype(np.int64) & 0xFFFFFFFF
    lo, hi = (au & 0xFFFF).astype(np.float64), (au >> 16).astype(np.float64)
    j = np.arange(N)
    tw = np.exp(1j * np.pi * j / N)
    fd = np.fft.fft(d * tw, axis=-1)

    def prod(x):
        y = np.fft.ifft(np.fft.fft(x * tw, axis=-1) * fd, axis=-1) * np.conj(tw)
        return np.rint(y.real).astype(np.int64)

    return wrap32(prod(lo) + (prod(hi) << 16))


def decompose(poly, l, log2_base):
    """tgsw.jl:99-117 on the host (used by MK key expansion): int32 [..., N] -> int32 [l, ..., N]."""
    c = np.asarray(poly, np.int32).astype(np.int64)
    offset = sum(1 << (32 - p * log2_base) for p in range(1, l + 1)) * (1 << (log2_base - 1))
    t = wrap32(c + offset).astype(np.int64)
    mask, half = (1 << log2_base) - 1, 1 << (log2_base - 1)
    return np.stack([(((t >> (32 - p * log2_base)) & mask) - half) for p in range(1, l + 1)]).astype(np.int32)
