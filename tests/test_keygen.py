"""Device-side cloud-key generation (tfhe_keygen_cloud_key): BootstrapKey = tgsw_encrypt of every LWE key
bit (bootstrap.jl:6-15, tgsw.jl:52-88, tlwe.jl:63-73) and KeyswitchKey (keyswitch.jl:14-41).  Every word of the
generated key is predicted from the documented Philox streams (tests/philox_ref.py) and the exact negacyclic products
of the host key generator; then the key is used: the HIP gates equal the oracle's words and decrypt correctly."""
import numpy as np
import pytest

import philox_ref as ph
from conftest import DEVICE_PAIRS


def test_philox4x32_10_known_answers():
    """Random123 known-answer vectors for philox4x32-10 (kat_vectors of the Random123 distribution)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = ph.philox4x32_10(*[np.array([c], np.uint32) for c in ctr], key[0], key[1])
        assert tuple(int(g[0]) for g in got) == want


def test_stream_helpers_shapes_and_moments():
    idx = np.arange(200000, dtype=np.uint64)
    seed = np.array([0x90ABCDEF, 0x12345678, 7, 8, 9, 10], np.uint32)
    w = ph.uniform_words(1, idx, seed)
    assert w.dtype == np.uint32 and abs(w.astype(np.float64).mean() / 2**32 - 0.5) < 0.01
    assert np.array_equal(w[:8], ph.uniform_words(1, idx[:8], seed))
    g = ph.gaussians(2, idx, seed)
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1.0) < 0.01 and np.abs(g).max() < 6.8
    # the two halves of the seed are independent: the mask words depend on words 0-1 only, the noise on words 2-5 only
    other_noise = seed.copy(); other_noise[2:] += 1
    other_mask = seed.copy(); other_mask[:2] += 1
    assert np.array_equal(ph.uniform_words(1, idx[:64], other_noise), w[:64])
    assert np.array_equal(ph.gaussians(2, idx[:64], other_mask), g[:64])
    for i in range(2, 6):
        one = seed.copy(); one[i] ^= 1
        assert not np.array_equal(ph.gaussians(2, idx[:64], one), g[:64]), f"noise does not depend on seed word {i}"


def _predict(tfhe, p, lwe_key, tlwe_key, seed):
    """The key the device must produce, from the documented streams: (bk, ks) as uint32, noise words within one unit."""
    from tfhe_jl_amd.numeric import negacyclic_mul_binary
    n, N, k, l, beta = p.lwe_size, p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length, p.bs_log2_base
    S = n * l * (k + 1)
    a = ph.uniform_words(1, np.arange(S * k * N, dtype=np.uint64), seed).reshape(S, k, N)
    noise = ph.dtot32(ph.gaussians(2, np.arange(S * N, dtype=np.uint64), seed) * p.bs_noise_stddev).reshape(S, N)
    body = noise.astype(np.int64)
    for c in range(k):
        body = body + negacyclic_mul_binary(tlwe_key[c], a[:, c, :].astype(np.int32)).astype(np.int64)
    bk = np.concatenate([a.astype(np.int64), body[:, None, :]], axis=1).reshape(n, l, k + 1, k + 1, N)
    for pp in range(l):
        for j in range(k + 1):
            bk[:, pp, j, j, 0] += lwe_key.astype(np.int64) << (32 - (pp + 1) * beta)
    t, lb = p.ks_decomp_length, p.ks_log2_base
    kN, b1 = k * N, (1 << lb) - 1
    Q = kN * t * b1
    ka = ph.uniform_words(3, np.arange(Q * n, dtype=np.uint64), seed).reshape(Q, n)
    g = ph.gaussians(4, np.arange(Q, dtype=np.uint64), seed) * p.ks_noise_stddev
    # fixed summation order of ks_mean_kernel: 256 strided partial sums, then a halving tree
    part = np.array([g[i::256].sum() if False else np.add.reduce(g[i::256]) for i in range(256)])
    h = 128
    while h:
        part[:h] = part[:h] + part[h:2 * h]
        h //= 2
    mean = part[0] / Q
    i = np.arange(kN)[:, None, None]; j = np.arange(1, t + 1)[None, :, None]; hh = np.arange(1, b1 + 1)[None, None, :]
    msg = ((tlwe_key.reshape(-1).astype(np.int64)[i] * hh) << (32 - j * lb)).reshape(Q)
    dot = ka.astype(np.int64) @ lwe_key.astype(np.int64)
    kb = msg + ph.dtot32(g - mean).astype(np.int64) + dot
    ks = np.concatenate([ka.astype(np.int64), kb[:, None]], axis=1).reshape(kN, t, b1, n + 1)
    return (bk & 0xFFFFFFFF).astype(np.uint32), (ks & 0xFFFFFFFF).astype(np.uint32)


def _close_mod32(got, want, tol):
    d = (got.astype(np.int64) - want.astype(np.int64) + 2**31) % 2**32 - 2**31
    return np.abs(d).max() <= tol, np.abs(d).max()


@pytest.mark.gpu
@pytest.mark.parametrize("n,N,k,l,beta", [(6, 1024, 1, 2, 10), (5, 1024, 2, 2, 10), (4, 2048, 1, 3, 7), (3, 1024, 1, 4, 6)])
def test_device_key_equals_prediction(tfhe, n, N, k, l, beta):
    base = tfhe.tfhe_parameters_80()
    p = tfhe.SchemeParameters(n, base.lwe_noise_stddev, N, k, l, beta, base.bs_noise_stddev, base.ks_decomp_length,
                              base.ks_log2_base, base.ks_noise_stddev, base.max_parties)
    rng = np.random.default_rng(1000 + n)
    lwe_key = rng.integers(0, 2, n).astype(np.int32)
    tlwe_key = rng.integers(0, 2, (k, N)).astype(np.int32)
    seed = np.array([0x01234567 + n, 0xC0FFEE, 0xDEADBEEF, 0x0BADF00D + n, 0x5EED5EED, 0x13572468], np.uint32)
    from tfhe_jl_amd import _lib
    eng = _lib.Engine(p, 0)
    bk, ks = eng.keygen_cloud_key(lwe_key, tlwe_key, p.bs_noise_stddev, p.ks_noise_stddev, seed)
    want_bk, want_ks = _predict(tfhe, p, lwe_key, tlwe_key, seed)
    gbk, gks = bk.view(np.uint32), ks.view(np.uint32)
    # mask words: exact.  bodies: exact up to the device's log / cos / sqrt differing from numpy's in the last place
    assert np.array_equal(gbk[:, :, :, :k, 1:], want_bk[:, :, :, :k, 1:]) and np.array_equal(gks[..., :n], want_ks[..., :n])
    ok, worst = _close_mod32(gbk, want_bk, 2)
    assert ok, worst
    ok, worst = _close_mod32(gks, want_ks, 2)
    assert ok, worst
    assert (gbk != want_bk).mean() < 0.01 and (gks != want_ks).mean() < 0.01
    # same seed -> same key; another seed -> another key
    bk2, ks2 = eng.keygen_cloud_key(lwe_key, tlwe_key, p.bs_noise_stddev, p.ks_noise_stddev, seed)
    assert np.array_equal(bk, bk2) and np.array_equal(ks, ks2)
    # another noise key: the published mask words stay, every body changes; another mask key: the masks change
    s2 = seed.copy(); s2[5] += 1
    bk3, _ = eng.keygen_cloud_key(lwe_key, tlwe_key, p.bs_noise_stddev, p.ks_noise_stddev, s2)
    assert np.array_equal(bk3[:, :, :, :k, 1:], bk[:, :, :, :k, 1:]) and not np.array_equal(bk3[:, :, :, k, :], bk[:, :, :, k, :])
    s3 = seed.copy(); s3[0] += 1
    bk4, _ = eng.keygen_cloud_key(lwe_key, tlwe_key, p.bs_noise_stddev, p.ks_noise_stddev, s3)
    assert not np.array_equal(bk4[:, :, :, :k, 1:], bk[:, :, :, :k, 1:])
    with pytest.raises(ValueError):
        eng.keygen_cloud_key(lwe_key, tlwe_key, p.bs_noise_stddev, p.ks_noise_stddev, seed[:2])
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["80", "128"])
def test_device_generated_key_runs_the_gates(tfhe, orc, which):
    """make_key_pair(keygen="device") at full size: the HIP gates on that key equal the oracle's words on the same key and
    decrypt to the truth table (runtests.jl:8-24)."""
    p = tfhe.tfhe_parameters_80() if which == "80" else tfhe.tfhe_parameters_128()
    rng = np.random.default_rng(77)
    sk, ck = tfhe.make_key_pair(rng, p, keygen="device")
    n, N, l = p.lwe_size, p.tlwe_polynomial_degree, p.bs_decomp_length
    assert ck.bootstrap_key.shape == (n, l, 2, 2, N) and ck.keyswitch_key.shape == (N, p.ks_decomp_length, 3, n + 1)
    eng = ck.engine(0)
    o = orc.Oracle(n, N, 1, l, p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    bx = np.array([0, 0, 1, 1] * 8, bool); by = np.array([0, 1, 0, 1] * 8, bool)
    x, y = tfhe.encrypt(rng, sk, bx).data, tfhe.encrypt(rng, sk, by).data
    for name, truth in (("NAND", ~(bx & by)), ("XOR", bx ^ by)):
        ops = np.full(bx.size, tfhe.OPCODES[name], np.uint8)
        got = eng.gates(ops, x, y)
        assert np.array_equal(tfhe.decrypt(sk, got), truth)
        assert np.array_equal(got[:6], o.gates(ops[:6], x[:6], y[:6]))
    ck.close()


@pytest.mark.gpu
def test_device_key_noise_statistics(tfhe):
    """Full-size 80-bit set: phase - message of every generated sample is Gaussian noise of the configured deviation
    (tlwe.jl:63-73 with bs_noise_stddev; keyswitch.jl:24-40 with ks_noise_stddev, mean removed)."""
    from tfhe_jl_amd import _lib
    from tfhe_jl_amd.numeric import negacyclic_mul_binary
    p = tfhe.tfhe_parameters_80()
    n, N, l, beta = p.lwe_size, p.tlwe_polynomial_degree, p.bs_decomp_length, p.bs_log2_base
    rng = np.random.default_rng(31)
    lwe_key = rng.integers(0, 2, n).astype(np.int32)
    tlwe_key = rng.integers(0, 2, (1, N)).astype(np.int32)
    eng = _lib.Engine(p, 0)
    bk, ks = eng.keygen_cloud_key(lwe_key, tlwe_key, p.bs_noise_stddev, p.ks_noise_stddev, np.arange(99, 105, dtype=np.uint32))
    eng.close()
    # bootstrap key: body - a * s - message
    a, b = bk[:, :, :, 0, :].astype(np.int64), bk[:, :, :, 1, :].astype(np.int64)
    msg = np.zeros_like(b)
    for pp in range(l):
        g = lwe_key.astype(np.int64) << (32 - (pp + 1) * beta)
        a[:, pp, 0, 0] -= g                       # row (p, 0): message on the mask's constant term
        msg[:, pp, 1, 0] = g                      # row (p, 1): on the body's
    prod = negacyclic_mul_binary(tlwe_key[0], (a & 0xFFFFFFFF).astype(np.uint32).view(np.int32)).astype(np.int64)
    e = ((b - prod - msg + 2**31) % 2**32 - 2**31) / 2.0**32
    assert abs(e.std() / p.bs_noise_stddev - 1.0) < 0.02 and abs(e.mean()) < 0.01 * p.bs_noise_stddev
    assert np.abs(e).max() < 7 * p.bs_noise_stddev
    # keyswitch key: b - <a, s> - message
    t, lb = p.ks_decomp_length, p.ks_log2_base
    kk = ks.astype(np.int64)
    i = np.arange(N)[:, None, None]; j = np.arange(1, t + 1)[None, :, None]; hh = np.arange(1, 4)[None, None, :]
    kmsg = (tlwe_key[0].astype(np.int64)[i] * hh) << (32 - j * lb)
    ke = ((kk[..., n] - kk[..., :n] @ lwe_key.astype(np.int64) - kmsg + 2**31) % 2**32 - 2**31) / 2.0**32
    assert abs(ke.std() / p.ks_noise_stddev - 1.0) < 0.02 and abs(ke.mean()) < 1e-3 * p.ks_noise_stddev   # mean removed
    # masks look uniform
    u = bk[:, :, :, 0, 1:].view(np.uint32).astype(np.float64) / 2**32
    assert abs(u.mean() - 0.5) < 1e-3 and abs(u.std() - 12 ** -0.5) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("devs", DEVICE_PAIRS)
def test_device_keygen_on_a_multi_device_context(tfhe, devs):
    """A fan-out context ({0, 0} on the one-GPU box; {0, 1}: the second device takes the key through a host copy) generates on its
    first device and replicates: same key, same gates."""
    p = tfhe.tfhe_parameters_80()
    rng = np.random.default_rng(5)
    sk = tfhe.SecretKey(rng, p)
    st = rng.bit_generator.state
    fixed = [11, 22, 33, 44]                 # (the noise key otherwise comes from os.urandom, whatever `rng` is)
    ck1 = tfhe.CloudKey(rng, sk, keygen="device", device=0, noise_seed=fixed)
    rng.bit_generator.state = st
    ck2 = tfhe.CloudKey(rng, sk, keygen="device", device=devs, noise_seed=fixed)
    rng.bit_generator.state = st
    ck3 = tfhe.CloudKey(rng, sk, keygen="device", device=0)
    assert not np.array_equal(ck1.bootstrap_key, ck3.bootstrap_key)               # a fresh noise key from the OS every time
    assert np.array_equal(ck1.bootstrap_key[..., 0, :], ck3.bootstrap_key[..., 0, :])   # ... under the same (public, rng-drawn) masks
    ck3.close()
    assert not hasattr(ck1, "keygen_seed") and sk.cloud_keygen_seed is not None      # the seed stays on the secret side
    assert np.array_equal(ck1.bootstrap_key, ck2.bootstrap_key) and np.array_equal(ck1.keyswitch_key, ck2.keyswitch_key)
    bits = rng.integers(0, 2, (2, 40)).astype(bool)
    x, y = tfhe.encrypt(rng, sk, bits[0]).data, tfhe.encrypt(rng, sk, bits[1]).data
    ops = np.zeros(40, np.uint8)
    a, b = ck1.engine(0).gates(ops, x, y), ck2.engine(devs).gates(ops, x, y)
    assert np.array_equal(a, b) and np.array_equal(tfhe.decrypt(sk, a), ~(bits[0] & bits[1]))
    ck1.close(); ck2.close()
