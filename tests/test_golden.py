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
    # every kernel variant gives the same words (br_small = -1: one wave per rotation even for small batches;
    # default: the two-waves-per-rotation kernel takes batches this small)
    for bv, kv, small in ((1, 1, -1), (2, 3, -1), (3, 3, -1), (2, 4, -1), (2, 4, 512)):
        e = tfhe.Engine(params, 0)
        e.set_option("ks_variant", kv)               # before the key load: only that family's layout is built
        e.load_bootstrap_key(kat["bootstrap_key"])
        e.load_keyswitch_key(kat["keyswitch_key"])
        e.set_option("br_variant", bv)
        e.set_option("br_small", small)
        assert np.array_equal(e.gates(kat["ops"], kat["in0"], kat["in1"], kat["in2"]), kat["out"])
        assert np.array_equal(e.keyswitch(kat["ext"]), kat["ks_out"])
        e.close()
