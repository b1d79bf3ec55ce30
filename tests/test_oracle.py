"""CPU tests of the oracle (oracle/tfhe_oracle.c) against what the reference pins:
closed-form constants, algebraic identities, and the truth tables of test/runtests.jl:8-57."""
import itertools

import numpy as np
import pytest

GATES = [  # test/runtests.jl:8-21
    ("NAND", 2, lambda x, y: not (x and y)),
    ("OR", 2, lambda x, y: x or y),
    ("AND", 2, lambda x, y: x and y),
    ("XOR", 2, lambda x, y: x != y),
    ("XNOR", 2, lambda x, y: x == y),
    ("NOT", 1, lambda x: not x),
    ("NOR", 2, lambda x, y: not (x or y)),
    ("ANDNY", 2, lambda x, y: (not x) and y),
    ("ANDYN", 2, lambda x, y: x and (not y)),
    ("ORNY", 2, lambda x, y: (not x) or y),
    ("ORYN", 2, lambda x, y: x or (not y)),
    ("MUX", 3, lambda x, y, z: y if x else z),
]


def test_encode_decode_constants(orc):
    # numeric-functions.jl:42-45 / SURVEY §8 a1
    assert orc.encode_message(1, 8) == 2**29
    assert orc.encode_message(-1, 8) == -2**29
    assert orc.encode_message(1, 4) == 2**30
    assert orc.encode_message(-1, 4) == -2**30
    # numeric-functions.jl:31-34: signed result in [-N, N); the wrapping add is intended
    assert orc.decode_message(0, 2048) == 0
    assert orc.decode_message(2**31 - 1, 2048) == -1024
    assert orc.decode_message(-2**31, 2048) == -1024
    assert orc.decode_message(2**21, 2048) == 1
    assert orc.decode_message(2**20, 2048) == 1
    assert orc.decode_message(2**20 - 1, 2048) == 0
    assert orc.decode_message(-2**20 - 1, 2048) == -1


def test_tgsw_constants(orc):
    # tgsw.jl:8-21: 80-bit set gadget {2^22, 2^12}, offset 0x80200000; 128-bit set gadget {2^25, 2^18, 2^11}
    g, off = orc.tgsw_constants(2, 10)
    assert list(g) == [2**22, 2**12]
    assert off & 0xFFFFFFFF == 0x80200000
    g, off = orc.tgsw_constants(3, 7)
    assert list(g) == [2**25, 2**18, 2**11]
    assert off & 0xFFFFFFFF == (2**31 + 2**24 + 2**17) & 0xFFFFFFFF


@pytest.mark.parametrize("l,beta", [(2, 10), (3, 7), (4, 7)])
def test_decompose_reconstructs(orc, l, beta):
    # tgsw.jl:91-97: digits in [-B/2, B/2); sum_p d_p 2^(32 - p beta) == c floored to a multiple of
    # 2^(32 - l beta) (the low bits are dropped, not rounded)
    rng = np.random.default_rng(5)
    c = rng.integers(-2**31, 2**31, size=1024, dtype=np.int64).astype(np.int32)
    c[:4] = [0, -1, 2**31 - 1, -2**31]
    d = orc.decompose(c, l, beta).astype(np.int64)
    assert d.min() >= -(1 << (beta - 1)) and d.max() < (1 << (beta - 1))
    rec = sum(d[p] << (32 - (p + 1) * beta) for p in range(l))
    err = ((rec - c.astype(np.int64) + 2**31) % 2**32) - 2**31
    assert err.max() <= 0 and err.min() > -(1 << (32 - l * beta))
    assert np.all(orc.decompose(np.zeros(16, np.int32), l, beta) == 0)   # decompose(0) == 0


def test_mul_by_monomial(orc):
    p = np.arange(1, 9, dtype=np.int32)
    assert list(orc.mul_by_monomial(p, 0)) == list(p)
    assert list(orc.mul_by_monomial(p, 1)) == [-8, 1, 2, 3, 4, 5, 6, 7]
    assert list(orc.mul_by_monomial(p, 8)) == list(-p)
    assert list(orc.mul_by_monomial(p, 16)) == list(p)
    assert list(orc.mul_by_monomial(p, -1)) == [2, 3, 4, 5, 6, 7, 8, -1]
    assert list(orc.mul_by_monomial(p, -9)) == list(orc.mul_by_monomial(p, 7))


@pytest.mark.parametrize("N,bits", [(1024, 10), (1024, 7), (2048, 10)])
def test_fft_product_equals_exact(orc, N, bits):
    """The reference's Float64 FFT product after rounding == exact negacyclic product mod 2^32."""
    rng = np.random.default_rng(N + bits)
    worst = 0.0
    for _ in range(6):
        a = rng.integers(-(1 << (bits - 1)), 1 << (bits - 1), size=N).astype(np.int32)
        b = rng.integers(-2**31, 2**31, size=N, dtype=np.int64).astype(np.int32)
        got, margin = orc.negacyclic_mul_fft(a, b)
        assert np.array_equal(got, orc.negacyclic_mul_exact(a, b))
        worst = max(worst, margin)
    assert worst < 0.25   # far from a flipped rounding (threshold 0.5)


def test_exact_product_small_case(orc):
    # (1 + 2X)(3 + 4X^3) mod X^4+1 = 3 + 6X + 4X^3 + 8X^4 -> 3 - 8 + 6X + 4X^3
    assert list(orc.negacyclic_mul_exact([1, 2, 0, 0], [3, 0, 0, 4])) == [-5, 6, 0, 4]


def test_forward_transform_is_odd_root_evaluation(orc):
    """polynomials.jl:106-112: spectrum k is conj(p(exp(i pi (4k+1)/N)))."""
    rng = np.random.default_rng(3)
    N = 1024
    p = rng.integers(-1000, 1000, size=N).astype(np.int32)
    spec = orc.forward_transform(p)
    j = np.arange(N)
    for k in (0, 1, 17, 511):
        root = np.exp(1j * np.pi * (4 * k + 1) / N)
        assert abs(np.conj(np.sum(p * root**j)) - spec[k]) < 1e-6


@pytest.mark.parametrize("name,nargs,ref", GATES, ids=[g[0] for g in GATES])
def test_truth_tables_80(orc, tfhe, keys80, name, nargs, ref):
    """test/runtests.jl:26-40 restated: decrypt(gate(encrypt...)) == reference(bits...)."""
    K = keys80
    combos = list(itertools.product((False, True), repeat=nargs))
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(nargs)]
    ops = np.full(len(combos), orc.OPS[name], np.uint8)
    out = K.oracle.gates(ops, *ins, nthreads=4)
    got = tfhe.decrypt(K.sk, out)
    assert list(got) == [bool(ref(*c)) for c in combos]
    assert K.oracle.last_margin < 0.25


def test_truth_table_128_nand(orc, tfhe, keys128):
    """test/runtests.jl:43-57"""
    K = keys128
    combos = list(itertools.product((False, True), repeat=2))
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(2)]
    out = K.oracle.gates(np.full(4, orc.OPS["NAND"], np.uint8), *ins, nthreads=4)
    assert list(tfhe.decrypt(K.sk, out)) == [not (a and b) for a, b in combos]


def test_fft_mode_equals_exact_mode_bootstrap(orc, tfhe, keys80):
    """Word-for-word agreement of the two product back-ends over a whole gate."""
    K = keys80
    x = tfhe.encrypt(K.rng, K.sk, [True]).data
    y = tfhe.encrypt(K.rng, K.sk, [False]).data
    ops = np.array([orc.OPS["NAND"]], np.uint8)
    a = K.oracle.gates(ops, x, y, mode=orc.MODE_FFT)
    b = K.oracle.gates(ops, x, y, mode=orc.MODE_EXACT)
    assert np.array_equal(a, b)


def test_constant_and_not(orc, tfhe, keys80):
    K = keys80
    x = tfhe.encrypt(K.rng, K.sk, [True, False]).data
    ops = np.array([orc.OPS["CONST1"], orc.OPS["CONST0"]], np.uint8)
    out = K.oracle.gates(ops, x)
    assert np.all(out[:, :-1] == 0) and out[0, -1] == 2**29 and out[1, -1] == -2**29
    out = K.oracle.gates(np.full(2, orc.OPS["NOT"], np.uint8), x)
    assert np.array_equal(out, (-x.astype(np.int64)).astype(np.int32))


def test_noiseless_bootstrap_kat(orc, tfhe):
    """Deterministic KAT: with a noiseless bootstrapping key the blind rotation of a trivial
    sample of phase +-1/4 extracts exactly (0, ..., 0, +-mu)."""
    n, N, l, beta = 8, 1024, 2, 10
    rng = np.random.default_rng(11)
    from tfhe_jl_amd.keys import TLweKey, make_bootstrap_key
    from tfhe_jl_amd.lwe import LweKey
    lwe_key = LweKey(rng, n)
    tlwe_key = TLweKey(rng, N, 1)
    bk = make_bootstrap_key(rng, 0.0, lwe_key, tlwe_key, l, beta)   # alpha = 0: noiseless
    o = orc.Oracle(n, N, 1, l, beta, 8, 2)
    o.load_bootstrap_key(bk)
    mu = 2**29
    for phase, want in ((2**30, mu), (-2**30, -mu)):
        x = np.zeros((1, n + 1), np.int32)
        x[0, -1] = phase
        ext = o.bootstrap(mu, x, with_keyswitch=False)
        # phase of the extracted sample under the extracted TLWE key
        s = tlwe_key.key.reshape(-1).astype(np.int64)
        ph = (int(ext[0, -1]) - int(ext[0, :-1].astype(np.int64) @ s)) % 2**32
        assert ph == want % 2**32


def test_truth_table_mask_size_2(orc, tfhe):
    """tfhe_parameters_80(tlwe_mask_size=2) (api.jl:30): NAND and MUX truth tables on the oracle."""
    from conftest import KeySet
    K = KeySet(tfhe, orc, tfhe.tfhe_parameters_80(tlwe_mask_size=2), seed=77)
    assert K.ck.bootstrap_key.shape == (500, 2, 3, 3, 1024) and K.ck.keyswitch_key.shape == (2048, 8, 3, 501)
    combos = list(itertools.product((False, True), repeat=3))
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(3)]
    out = K.oracle.gates(np.full(8, orc.OPS["MUX"], np.uint8), *ins, nthreads=8)
    assert list(tfhe.decrypt(K.sk, out)) == [bool(y if x else z) for x, y, z in combos]
    out = K.oracle.gates(np.full(8, orc.OPS["NAND"], np.uint8), *ins, nthreads=8)
    assert list(tfhe.decrypt(K.sk, out)) == [not (x and y) for x, y, z in combos]


def synthetic_2048(tfhe, n=630):
    """BASELINE config 4b: SchemeParameters(630, 2^-15, 2048, 1, 3, 7, 2^-25, 8, 2, 2^-15, 1) — synthetic (the
    reference ships no N = 2048 set; SchemeParameters is an unvalidated positional struct, api.jl:4-21)."""
    return tfhe.SchemeParameters(n, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)


def test_truth_table_synthetic_n2048(orc, tfhe):
    from conftest import KeySet
    K = KeySet(tfhe, orc, synthetic_2048(tfhe, n=96), seed=2048)
    combos = list(itertools.product((False, True), repeat=3))
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(3)]
    out = K.oracle.gates(np.full(8, orc.OPS["MUX"], np.uint8), *ins, nthreads=8)
    assert list(tfhe.decrypt(K.sk, out)) == [bool(y if x else z) for x, y, z in combos]
    assert K.oracle.last_margin < 0.25
    one = K.oracle.gates(np.array([orc.OPS["NAND"]], np.uint8), ins[0][:1], ins[1][:1])
    assert np.array_equal(one, K.oracle.gates(np.array([orc.OPS["NAND"]], np.uint8), ins[0][:1], ins[1][:1], mode=orc.MODE_EXACT))
