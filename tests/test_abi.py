"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/tfhe_mi355x.h declares; host-side argument validation; no compute calls (no GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "tfhe_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(tfhe_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(tfhe):
    lib = tfhe._lib.load()
    syms = _header_symbols()
    assert len(syms) >= 19
    for s in syms:
        assert hasattr(lib, s), f"libtfhe_mi355x.so does not export {s}"
    assert sorted(tfhe._lib.ABI_SYMBOLS) == syms
    assert lib.tfhe_abi_version() == 7      # (a development build would answer -7 and tfhe._lib.load() would have refused it)


def test_opcode_numbering_matches_header(tfhe, orc):
    txt = open(os.path.join(ROOT, "include", "tfhe_mi355x.h")).read()
    for name, val in tfhe.OPCODES.items():
        m = re.search(rf"TFHE_GATE_{name}\s*=\s*(\d+)", txt)
        assert m and int(m.group(1)) == val
        assert orc.OPS[name] == val


def test_ctx_create_validates_parameters(tfhe):
    lib = tfhe._lib.load()
    P = tfhe._lib.TfheParams
    h = C.c_void_p()
    # N not a power of two / l*beta > 32 / t*gamma > 31: rejected before any device work
    for bad in (P(500, 1000, 1, 2, 10, 8, 2, 1), P(500, 1024, 1, 4, 10, 8, 2, 1), P(500, 1024, 1, 2, 10, 16, 2, 1)):
        rc = lib.tfhe_ctx_create(C.byref(bad), 0, C.byref(h))
        assert rc == 1 and not h.value
        assert b"tfhe_ctx_create" in lib.tfhe_last_error(None)
    assert lib.tfhe_ctx_create(None, 0, C.byref(h)) == 1


def test_engine_fails_loudly_without_device(tfhe):
    """No silent CPU fallback: on a box without a GPU creating a context is an error."""
    lib = tfhe._lib.load()
    if lib.tfhe_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(tfhe.EngineError):
        tfhe.Engine(tfhe.tfhe_parameters_80())


def test_host_keygen_shapes_and_roundtrip(tfhe, keys80):
    K = keys80
    p = K.params
    assert K.ck.bootstrap_key.shape == (p.lwe_size, p.bs_decomp_length, 2, 2, p.tlwe_polynomial_degree)
    assert K.ck.keyswitch_key.shape == (1024, 8, 3, 501)
    bits = K.rng.integers(0, 2, 64).astype(bool)
    assert np.array_equal(tfhe.decrypt(K.sk, tfhe.encrypt(K.rng, K.sk, bits)), bits)
    s = tfhe.encrypt(K.rng, K.sk, True)
    assert isinstance(s, tfhe.LweSample) and tfhe.decrypt(K.sk, s) is True


def test_cloud_key_file_roundtrip(tfhe, tmp_path):
    """The engine's flat key container (the reference has no serialisation)."""
    params = tfhe.SchemeParameters(8, 1 / 2**15, 1024, 1, 2, 10, 9e-9, 8, 2, 1 / 2**15, 1)
    sk, ck = tfhe.make_key_pair(np.random.default_rng(5), params)
    path = str(tmp_path / "ck.tfhe")
    tfhe.save_cloud_key(path, ck)
    back = tfhe.load_cloud_key(path)
    assert back.params == params
    assert np.array_equal(back.bootstrap_key, ck.bootstrap_key) and np.array_equal(back.keyswitch_key, ck.keyswitch_key)
    with open(path, "r+b") as f:
        f.write(b"X")
    with pytest.raises(ValueError):
        tfhe.load_cloud_key(path)
