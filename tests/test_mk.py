"""Multi-key (2-party) NAND — BASELINE config 5, reference test "multikey NAND" (test/runtests.jl:60-100).
CPU: the oracle's MK restatement decrypts correctly under the host MK key generator and its two product
back-ends agree.  GPU: the HIP path equals the oracle word for word (decrypt-level MK tests are inherently
~0.2 %/gate flaky — SURVEY §4 — so parity is asserted on ciphertext words)."""
import numpy as np
import pytest


class MKKeys:
    def __init__(self, tfhe, orc, n=None, seed=321):
        p = tfhe.mktfhe_parameters_2party
        if n is not None:   # reduced lwe_size for fast CPU tests; everything else as mk_api.jl:4-10
            p = tfhe.SchemeParameters(n, p.lwe_noise_stddev, 1024, 1, p.bs_decomp_length, p.bs_log2_base,
                                      p.bs_noise_stddev, p.ks_decomp_length, p.ks_log2_base, p.ks_noise_stddev, 2)
        self.params = p
        self.rng = np.random.default_rng(seed)
        self.sks = [tfhe.SecretKey(self.rng, p) for _ in range(2)]                  # runtests.jl:69
        shared = tfhe.SharedKey(self.rng, p)                                        # :72
        parts = [tfhe.CloudKeyPart(self.rng, sk, shared) for sk in self.sks]        # :75
        self.ck = tfhe.MKCloudKey(parts)                                            # :79
        self.oracle = orc.Oracle(p.lwe_size, 1024, 1, p.bs_decomp_length, p.bs_log2_base, p.ks_decomp_length,
                                 p.ks_log2_base, parties=2)
        self.oracle.load_bootstrap_key(self.ck.bootstrap_key)
        self.oracle.load_keyswitch_key(self.ck.keyswitch_key)


@pytest.fixture(scope="module")
def mk_small(tfhe, orc):
    return MKKeys(tfhe, orc, n=24)


@pytest.fixture(scope="module")
def mk_full(tfhe, orc):
    return MKKeys(tfhe, orc)


def test_mk_encrypt_decrypt_roundtrip(tfhe, mk_small):
    K = mk_small
    bits = K.rng.integers(0, 2, 32).astype(bool)
    assert np.array_equal(tfhe.mk_decrypt(K.sks, tfhe.mk_encrypt(K.rng, K.sks, bits)), bits)   # runtests.jl:90-93
    assert tfhe.mk_decrypt(K.sks, tfhe.mk_encrypt(K.rng, K.sks, True)) is True


def test_mk_oracle_backends_agree(orc, tfhe, mk_small):
    K = mk_small
    x = tfhe.mk_encrypt(K.rng, K.sks, [True, False])
    y = tfhe.mk_encrypt(K.rng, K.sks, [True, True])
    a = K.oracle.mk_gate_nand(x, y, mode=orc.MODE_FFT)
    assert K.oracle.last_margin < 0.25
    assert np.array_equal(a, K.oracle.mk_gate_nand(x, y, mode=orc.MODE_EXACT))


def test_mk_oracle_nand_decrypts(orc, tfhe, mk_full):
    """test/runtests.jl:82-99 on the oracle: 10 random trials."""
    K = mk_full
    m1, m2 = K.rng.integers(0, 2, 10).astype(bool), K.rng.integers(0, 2, 10).astype(bool)
    out = K.oracle.mk_gate_nand(tfhe.mk_encrypt(K.rng, K.sks, m1), tfhe.mk_encrypt(K.rng, K.sks, m2), nthreads=8)
    ok = tfhe.mk_decrypt(K.sks, out) == ~(m1 & m2)
    assert ok.sum() >= 9      # ~3 sigma margin per gate: allow one noise failure in ten


@pytest.mark.gpu
def test_mk_gpu_parity(orc, tfhe, mk_full):
    K = mk_full
    B = 24
    m1, m2 = K.rng.integers(0, 2, B).astype(bool), K.rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.mk_encrypt(K.rng, K.sks, m1), tfhe.mk_encrypt(K.rng, K.sks, m2)
    got = tfhe.mk_gate_nand(K.ck, x, y)
    want = K.oracle.mk_gate_nand(x, y, nthreads=8)
    assert np.array_equal(got, want)
    assert (tfhe.mk_decrypt(K.sks, got) == ~(m1 & m2)).sum() >= B - 2
    # arbitrary input words (mod-switch edges) and a single sample
    z = K.rng.integers(-2**31, 2**31, size=(3, 1001), dtype=np.int64).astype(np.int32)
    z[0, :4] = [2**31 - 1, -2**31, 2**20, 0]
    assert np.array_equal(tfhe.mk_gate_nand(K.ck, z, z[::-1].copy()), K.oracle.mk_gate_nand(z, z[::-1].copy(), nthreads=4))
    one = tfhe.mk_gate_nand(K.ck, x[0], y[0])
    assert one.shape == (1001,) and np.array_equal(one, want[0])
    K.ck.close()


def _mk_setup(tfhe, orc, base, parties, n, seed):
    p = tfhe.SchemeParameters(n, base.lwe_noise_stddev, 1024, 1, base.bs_decomp_length, base.bs_log2_base,
                              base.bs_noise_stddev, base.ks_decomp_length, base.ks_log2_base, base.ks_noise_stddev,
                              base.max_parties)
    rng = np.random.default_rng(seed)
    sks = [tfhe.SecretKey(rng, p) for _ in range(parties)]
    shared = tfhe.SharedKey(rng, p)
    ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, sk, shared) for sk in sks])
    o = orc.Oracle(n, 1024, 1, p.bs_decomp_length, p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base, parties=parties)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    return p, rng, sks, ck, o


@pytest.mark.parametrize("which,parties,n", [("4party", 4, 12), ("4party", 3, 12), ("8party", 8, 6)])
def test_mk_oracle_backends_agree_many_parties(orc, tfhe, which, parties, n):
    """mktfhe_parameters_4party / _8party (mk_api.jl:16-34; reduced lwe_size): FFT and exact back-ends agree."""
    base = getattr(tfhe, "mktfhe_parameters_" + which)
    p, rng, sks, ck, o = _mk_setup(tfhe, orc, base, parties, n, seed=40 + parties)
    assert ck.bootstrap_key.shape == (parties, n, 2 * p.bs_decomp_length * parties + 2 * p.bs_decomp_length, 1024)
    x = tfhe.mk_encrypt(rng, sks, [True, False])
    y = tfhe.mk_encrypt(rng, sks, [True, True])
    a = o.mk_gate_nand(x, y, mode=orc.MODE_FFT)
    assert o.last_margin < 0.25
    assert np.array_equal(a[:1], o.mk_gate_nand(x[:1], y[:1], mode=orc.MODE_EXACT))


@pytest.mark.gpu
@pytest.mark.parametrize("which,parties,n", [("2party", 2, 24), ("4party", 4, 12), ("4party", 3, 12), ("8party", 8, 6)])
def test_mk_gpu_parity_many_parties(orc, tfhe, which, parties, n):
    """The any-P multi-key kernel against the oracle, word for word; for 2 parties also against
    the specialised 2-party kernel."""
    base = getattr(tfhe, "mktfhe_parameters_" + which)
    p, rng, sks, ck, o = _mk_setup(tfhe, orc, base, parties, n, seed=40 + parties)
    B = 5
    x = rng.integers(-2**31, 2**31, size=(B, parties * n + 1), dtype=np.int64).astype(np.int32)
    y = rng.integers(-2**31, 2**31, size=(B, parties * n + 1), dtype=np.int64).astype(np.int32)
    x[:2] = tfhe.mk_encrypt(rng, sks, [True, False])
    y[:2] = tfhe.mk_encrypt(rng, sks, [True, True])
    eng = ck.engine(0)
    if parties == 2:
        special = eng.mk_gate_nand(x, y)
        eng.set_option("mk_general", 1)
    want = o.mk_gate_nand(x, y, nthreads=5)
    got = eng.mk_gate_nand(x, y)
    assert np.array_equal(got, want)
    if (parties, p.bs_decomp_length) in ((4, 5), (8, 8)):
        # the shipped 4- / 8-party shapes take the two-wave kernel with compile-time (parties, l) ...
        assert eng.last_kernel_name() == f"mk_blind_rotate_kernel_g2<{parties},{p.bs_decomp_length}" + (",acc=lds>" if parties == 4 else ">")
        assert np.array_equal(eng.mk_gate_nand(x[:1], y[:1]), want[:1])  # ... a single rotation with its padding partner
        eng.set_option("mkg_variant", 1)                                  # ... and the any-party kernel stays the fallback
        got = eng.mk_gate_nand(x, y)
        assert np.array_equal(got, want)
    assert eng.last_kernel_name().startswith("mk_blind_rotate_kernel_general")
    assert ("acc=global" in eng.last_kernel_name()) == (parties > 4)      # default placement of the accumulators
    if parties == 2:
        assert np.array_equal(got, special)
    # both placements of the accumulator polynomials (LDS / global memory), with and without lockstep groups
    for acc, rw in ((1, 0), (0, 0), (1, 1), (0, 1)):
        eng.set_option("mkg_acc", acc)
        eng.set_option("mkg_rw", rw)
        assert np.array_equal(eng.mk_gate_nand(x, y), want), (acc, rw)
        assert ("acc=global" in eng.last_kernel_name()) == bool(acc)
    ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize("which,parties,n", [("2party", 2, 12), ("4party", 4, 8), ("4party", 3, 8), ("8party", 8, 4)])
def test_device_rgsw_expand_equals_host(orc, tfhe, which, parties, n):
    """RGSW.Expand on the GPU (tfhe_mk_expand_load_bootstrap_key; mk_internals.jl:304-345): the expanded Int32 key equals
    the host expansion word for word, and the engine it leaves loaded computes the same NAND words as the oracle."""
    base = getattr(tfhe, "mktfhe_parameters_" + which)
    p = tfhe.SchemeParameters(n, base.lwe_noise_stddev, 1024, 1, base.bs_decomp_length, base.bs_log2_base, base.bs_noise_stddev,
                              base.ks_decomp_length, base.ks_log2_base, base.ks_noise_stddev, base.max_parties)
    rng = np.random.default_rng(900 + parties)
    sks = [tfhe.SecretKey(rng, p) for _ in range(parties)]
    shared = tfhe.SharedKey(rng, p)
    parts = [tfhe.CloudKeyPart(rng, sk, shared) for sk in sks]
    host = tfhe.MKCloudKey(parts)                        # numpy expansion
    dev = tfhe.MKCloudKey(parts, expand="device")
    assert np.array_equal(dev.bootstrap_key, host.bootstrap_key)
    o = orc.Oracle(n, 1024, 1, p.bs_decomp_length, p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base, parties=parties)
    o.load_bootstrap_key(host.bootstrap_key)
    o.load_keyswitch_key(host.keyswitch_key)
    x = tfhe.mk_encrypt(rng, sks, [True, False, True])
    y = tfhe.mk_encrypt(rng, sks, [True, True, False])
    assert np.array_equal(tfhe.mk_gate_nand(dev, x, y), o.mk_gate_nand(x, y, nthreads=3))
    host.close()
    dev.close()


@pytest.mark.gpu
def test_mk_four_party_full_size_sample(orc, tfhe):
    """mktfhe_parameters_4party at its shipped size (mk_api.jl:16-22: n = 500, l = 5, beta = 6): the key is expanded on
    the device (RGSW.Expand), downloaded for the oracle, and a sample of NAND gates is compared word for word; every
    output decrypts (2000 CMUX steps, 30 forward transforms each, through mk_blind_rotate_kernel_g2<4,5,acc=lds>)."""
    p = tfhe.mktfhe_parameters_4party
    rng = np.random.default_rng(4444)
    sks = [tfhe.SecretKey(rng, p) for _ in range(4)]
    shared = tfhe.SharedKey(rng, p)
    ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, sk, shared) for sk in sks], expand="device")
    eng = ck.engine(0)
    B = 64
    m1, m2 = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.mk_encrypt(rng, sks, m1), tfhe.mk_encrypt(rng, sks, m2)
    got = eng.mk_gate_nand(x, y)
    assert eng.last_kernel_name() == "mk_blind_rotate_kernel_g2<4,5,acc=lds>"
    assert (tfhe.mk_decrypt(sks, got) == ~(m1 & m2)).mean() >= 0.95
    o = orc.Oracle(p.lwe_size, 1024, 1, p.bs_decomp_length, p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base, parties=4)
    o.load_bootstrap_key(ck.bootstrap_key)          # the device-expanded key, downloaded
    o.load_keyswitch_key(ck.keyswitch_key)
    idx = [0, 1, 31, 63]
    assert np.array_equal(got[idx], o.mk_gate_nand(x[idx], y[idx], nthreads=4))
    ck.close()
