"""Committed known-answer vectors (tests/golden/kat_n4.npz, minted by tests/golden/gen_golden.py from the
oracle — restatement-derived, see that script's header).  CPU: the oracle still reproduces them (both
back-ends).  GPU: the HIP path reproduces them through the C ABI."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kat_n4.npz")


@pytest.fixture(scope="module")
def kat():
    return np.load(GOLD)


def _oracle(orc, kat):
    n, N, k, l, beta, t, g = [int(v) for v in kat["params"]]
    o = orc.Oracle(n, N, k, l, beta, t, g)
    o.load_bootstrap_key(kat["bootstrap_key"])
    o.load_keyswitch_key(kat["keyswitch_key"])
    return o


def test_oracle_reproduces_golden(orc, kat):
    o = _oracle(orc, kat)
    for mode in (orc.MODE_FFT, orc.MODE_EXACT):
        assert np.array_equal(o.gates(kat["ops"], kat["in0"], kat["in1"], kat["in2"], mode=mode), kat["out"])
    assert np.array_equal(o.bootstrap(2**29, kat["in0"][:8], with_keyswitch=False), kat["ext"])
    assert np.array_equal(o.keyswitch(kat["ext"]), kat["ks_out"])


@pytest.mark.gpu
def test_gpu_reproduces_golden(tfhe, kat):
    n, N, k, l, beta, t, g = [int(v) for v in kat["params"]]
    params = tfhe.SchemeParameters(n, 0.0, N, k, l, beta, 0.0, t, g, 0.0, 1)
    e = tfhe.Engine(params, 0)
    e.load_bootstrap_key(kat["bootstrap_key"])
    e.load_keyswitch_key(kat["keyswitch_key"])
    assert np.array_equal(e.gates(kat["ops"], kat["in0"], kat["in1"], kat["in2"]), kat["out"])
    assert np.array_equal(e.bootstrap(2**29, kat["in0"][:8], with_keyswitch=False), kat["ext"])
    assert np.array_equal(e.keyswitch(kat["ext"]), kat["ks_out"])
    # the keyswitch family decides which key layout is resident: it cannot change under a loaded key
    with pytest.raises(tfhe.EngineError) as ei:
        e.set_option("ks_variant", 3)
    assert ei.value.code == 5
    e.close()
    # every kernel family gives the same words (br_small = -1 / br_tiny = -1: one wave per rotation even for small batches;
    # default: the multi-wave kernels take batches this small; br_general: the any-parameter kernel)
    for bg, kv, small in ((0, 1, -1), (0, 3, -1), (1, 3, -1), (1, 4, -1), (0, 4, -1), (0, 4, 512)):
        e = tfhe.Engine(params, 0)
        e.set_option("ks_variant", kv)               # before the key load: only that family's layout is built
        e.load_bootstrap_key(kat["bootstrap_key"])
        e.load_keyswitch_key(kat["keyswitch_key"])
        e.set_option("br_general", bg)
        e.set_option("br_small", small)
        e.set_option("br_tiny", -1 if small < 0 else -2)
        assert np.array_equal(e.gates(kat["ops"], kat["in0"], kat["in1"], kat["in2"]), kat["out"])
        assert np.array_equal(e.keyswitch(kat["ext"]), kat["ks_out"])
        e.close()


# ---- fixtures minted by the REAL reference (julia/TFHEMI355X/scripts/mint_fixtures.jl) ------------------------------------------------
# Any tests/golden/ref_*.tfhe present is checked word for word against the oracle (CPU) and the HIP engine (GPU).
# None can be minted in the build image (no Julia): the tests then skip, and a Python twin of the Julia writer — same
# container, same section names, data from this repo's own keygen + oracle — keeps the reader / consumer code exercised.
import glob

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _params_of(sec, tfhe):
    n, N, k, l, b, t, g, parties = [int(v) for v in sec["params"]]
    nz = sec.get("noise", np.zeros(3))
    return tfhe.SchemeParameters(n, float(nz[0]), N, k, l, b, float(nz[1]), t, g, float(nz[2]), parties)


def check_reference_fixture(sec, tfhe, orc, gpu):
    """sec: sections of one fixture file.  Asserts that the oracle's (gpu=False) or the engine's (gpu=True) output
    words equal the fixture's `out`, and that `out` decrypts to `plain` under the fixture's secret key(s)."""
    p = _params_of(sec, tfhe)
    if "mk_spectra" in sec:
        P = int(sec["parties"][0])
        assert sec["mk_spectra"].shape == (P, p.lwe_size, 2 * p.bs_decomp_length * P + 2 * p.bs_decomp_length, p.tlwe_polynomial_degree // 2)
        a = sec["in0"][:, :-1].reshape(-1, P, p.lwe_size).astype(np.int64)       # phase = b - sum_p <a_p, s_p>  (mk_internals.jl:29-35)
        ph = (sec["out"][:, -1].astype(np.int64) - np.einsum("bpn,pn->b", sec["out"][:, :-1].reshape(-1, P, p.lwe_size).astype(np.int64),
                                                              sec["lwe_keys"].astype(np.int64))).astype(np.int32)
        assert np.array_equal((ph > 0).astype(np.uint8), sec["plain"]) and a.shape[0] == sec["out"].shape[0]
        if gpu:
            e = tfhe.Engine(p, 0)
            e.mk_load_bootstrap_key_spectra(sec["mk_spectra"], P)
            e.mk_load_keyswitch_key(sec["mk_keyswitch_key"], P)
            got = e.mk_gate_nand(sec["in0"], sec["in1"])
            e.close()
        else:
            o = orc.Oracle(p.lwe_size, p.tlwe_polynomial_degree, 1, p.bs_decomp_length, p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base, parties=P)
            o.load_bootstrap_spectra(sec["mk_spectra"])
            o.load_keyswitch_key(sec["mk_keyswitch_key"])
            got = o.mk_gate_nand(sec["in0"], sec["in1"], nthreads=8)
            assert np.array_equal(got, o.mk_gate_nand(sec["in0"], sec["in1"], mode=orc.MODE_EXACT, nthreads=8))
    else:
        k1 = p.tlwe_mask_size + 1
        assert sec["bk_spectra"].shape == (p.lwe_size, p.bs_decomp_length, k1, k1, p.tlwe_polynomial_degree // 2)
        ph = (sec["out"][:, -1].astype(np.int64) - sec["out"][:, :-1].astype(np.int64) @ sec["lwe_key"].astype(np.int64)).astype(np.int32)
        assert np.array_equal((ph > 0).astype(np.uint8), sec["plain"])                 # lwe.jl:59, api.jl:167-169
        if gpu:
            e = tfhe.Engine(p, 0)
            e.load_bootstrap_key_spectra(sec["bk_spectra"])
            e.load_keyswitch_key(sec["keyswitch_key"])
            got = e.gates(sec["ops"], sec["in0"], sec["in1"], sec["in2"])
            e.close()
        else:
            o = orc.Oracle(p.lwe_size, p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length, p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base)
            o.load_bootstrap_spectra(sec["bk_spectra"])
            o.load_keyswitch_key(sec["keyswitch_key"])
            got = o.gates(sec["ops"], sec["in0"], sec["in1"], sec["in2"], nthreads=8)
            assert np.array_equal(got, o.gates(sec["ops"], sec["in0"], sec["in1"], sec["in2"], mode=orc.MODE_EXACT, nthreads=8))
    assert np.array_equal(got, sec["out"]), "output words differ from the fixture's"


def _reference_files():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, "ref_*.tfhe")))


def test_oracle_reproduces_reference_fixtures(tfhe, orc):
    files = _reference_files()
    if not files:
        pytest.skip("no tests/golden/ref_*.tfhe: mint them with julia/TFHEMI355X/scripts/mint_fixtures.jl (needs Julia + TFHE.jl)")
    from tfhe_jl_amd.serialize import read_sections
    for f in files:
        check_reference_fixture(read_sections(f), tfhe, orc, gpu=False)


@pytest.mark.gpu
def test_gpu_reproduces_reference_fixtures(tfhe, orc):
    files = _reference_files()
    if not files:
        pytest.skip("no tests/golden/ref_*.tfhe: mint them with julia/TFHEMI355X/scripts/mint_fixtures.jl (needs Julia + TFHE.jl)")
    from tfhe_jl_amd.serialize import read_sections
    for f in files:
        check_reference_fixture(read_sections(f), tfhe, orc, gpu=True)


def _mint_twin(tfhe, orc, path, multikey, N=1024):
    """Python twin of julia/TFHEMI355X/scripts/mint_fixtures.jl (same container and section names), data from this repo's keygen + oracle.
    N = 512: the twin of ref_gates_n512.tfhe / ref_mk2_n512.tfhe (a set the engine runs on its any-N kernels)."""
    from tfhe_jl_amd.serialize import write_sections
    rng = np.random.default_rng(5)
    if multikey:
        b = tfhe.mktfhe_parameters_2party
        p = tfhe.SchemeParameters(8, b.lwe_noise_stddev, N, 1, 4, 7, b.bs_noise_stddev, 8, 2, b.ks_noise_stddev, 2)
        sks = [tfhe.SecretKey(rng, p) for _ in range(2)]
        shared = tfhe.SharedKey(rng, p)
        ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, s, shared) for s in sks])
        o = orc.Oracle(8, N, 1, 4, 7, 8, 2, parties=2)
        o.load_bootstrap_key(ck.bootstrap_key)
        o.load_keyswitch_key(ck.keyswitch_key)
        bits = rng.integers(0, 2, (2, 6)).astype(bool)
        x, y = tfhe.mk_encrypt(rng, sks, bits[0]), tfhe.mk_encrypt(rng, sks, bits[1])
        out = o.mk_gate_nand(x, y)
        write_sections(path, {
            "params": np.array(p.engine_tuple(), np.int32), "noise": np.array([p.lwe_noise_stddev, p.bs_noise_stddev, p.ks_noise_stddev]),
            "parties": np.array([2], np.int32), "mk_spectra": o.bk_spectra(), "mk_keyswitch_key": ck.keyswitch_key,
            "lwe_keys": np.stack([s.key.key for s in sks]).astype(np.int32), "in0": x, "in1": y, "out": out,
            "plain": tfhe.mk_decrypt(sks, out).astype(np.uint8)})
    else:
        p = tfhe.SchemeParameters(8, 1 / 2**15, N, 1, 2, 10, 9e-9, 8, 2, 1 / 2**15, 1)
        sk, ck = tfhe.make_key_pair(rng, p)
        o = orc.Oracle(8, N, 1, 2, 10, 8, 2)
        o.load_bootstrap_key(ck.bootstrap_key)
        o.load_keyswitch_key(ck.keyswitch_key)
        names = ["NAND", "OR", "AND", "XOR", "XNOR", "NOT", "NOR", "ANDNY", "ANDYN", "ORNY", "ORYN", "MUX", "CONST0", "CONST1"]
        ops = np.repeat(np.array([tfhe.OPCODES[n] for n in names], np.uint8), 2)
        ins = [tfhe.encrypt(rng, sk, rng.integers(0, 2, ops.size).astype(bool)).data for _ in range(3)]
        out = o.gates(ops, *ins)
        write_sections(path, {
            "params": np.array(p.engine_tuple(), np.int32), "noise": np.array([p.lwe_noise_stddev, p.bs_noise_stddev, p.ks_noise_stddev]),
            "bk_spectra": o.bk_spectra(), "keyswitch_key": ck.keyswitch_key, "lwe_key": sk.key.key.astype(np.int32),
            "ops": ops, "in0": ins[0], "in1": ins[1], "in2": ins[2], "out": out, "plain": tfhe.decrypt(sk, out).astype(np.uint8)})


@pytest.mark.parametrize("N", [1024, 512])
@pytest.mark.parametrize("multikey", [False, True], ids=["single-key", "multi-key"])
def test_fixture_reader_on_python_twin(tfhe, orc, tmp_path, multikey, N):
    from tfhe_jl_amd.serialize import read_sections
    f = str(tmp_path / "twin.tfhe")
    _mint_twin(tfhe, orc, f, multikey, N)
    check_reference_fixture(read_sections(f), tfhe, orc, gpu=False)


@pytest.mark.gpu
@pytest.mark.parametrize("N", [1024, 512])
@pytest.mark.parametrize("multikey", [False, True], ids=["single-key", "multi-key"])
def test_fixture_reader_on_python_twin_gpu(tfhe, orc, tmp_path, multikey, N):
    from tfhe_jl_amd.serialize import read_sections
    f = str(tmp_path / "twin.tfhe")
    _mint_twin(tfhe, orc, f, multikey, N)
    check_reference_fixture(read_sections(f), tfhe, orc, gpu=True)
