"""numpy restatement of the key generator's random streams (tfhe.jl_amd/csrc/kernels_keygen.hpp): Philox4x32-10
(Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), the uniform-word and Gaussian
conventions, and the stream numbering.  Test infrastructure: lets the tests predict every word of a device-generated key."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over the counter words (uint32 arrays of one shape); k0, k1: Python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, np.uint64) & MASK for c in np.broadcast_arrays(c0, c1, c2, c3))
    for r in range(10):
        p0, p1 = M0 * c0, M1 * c2
        kk0, kk1 = np.uint64((k0 + r * W0) & 0xFFFFFFFF), np.uint64((k1 + r * W1) & 0xFFFFFFFF)
        c0, c1, c2, c3 = (p1 >> np.uint64(32)) ^ c1 ^ kk0, p1 & MASK, (p0 >> np.uint64(32)) ^ c3 ^ kk1, p0 & MASK
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def uniform_words(stream, index, seed):
    """Word `index` (uint64 array) of uniform (mask) stream `stream`: output (index & 3) of block (index >> 2), keyed by
    seed[0..1] (seed: six 32-bit words, kernels_keygen.hpp)."""
    index = np.asarray(index, np.uint64)
    blk = index >> np.uint64(2)
    out = philox4x32_10(blk & MASK, blk >> np.uint64(32), np.uint64(stream), np.uint64(0), int(seed[0]), int(seed[1]))
    sel = (index & np.uint64(3)).astype(np.int64)
    return np.choose(sel, out).astype(np.uint32)


def gaussians(stream, index, seed):
    """Standard normal `index` of Gaussian (noise) stream `stream`: Box-Muller on outputs 0, 1 of block `index`, keyed by the
    secret half of the seed: Philox key seed[2..3], counter words 2, 3 = stream ^ seed[4], seed[5]."""
    index = np.asarray(index, np.uint64)
    x, y, _, _ = philox4x32_10(index & MASK, index >> np.uint64(32), np.uint64(stream ^ int(seed[4])), np.uint64(int(seed[5])),
                               int(seed[2]), int(seed[3]))
    u1 = (x.astype(np.float64) + 0.5) / 4294967296.0
    u2 = (y.astype(np.float64) + 0.5) / 4294967296.0
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(6.283185307179586476925 * u2)


def dtot32(d):
    """trunc(Int32, d * 2^32), wrapping (numeric-functions.jl:51-53)."""
    return np.trunc(np.asarray(d, np.float64) * 4294967296.0).astype(np.int64).astype(np.uint32)
