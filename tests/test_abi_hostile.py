"""Hostile arguments at the raw C ABI (ctypes, no Python wrapper in between): NULL contexts and pointers, negative and absurd
counts, indices outside the wire table, multi-key calls on a single-key context, unknown options — every call must come back
with a non-zero status and a message, none may crash or corrupt the context: the same context computes a correct gate at the end.
(The reference has no validation at all — SchemeParameters is an unvalidated positional struct, api.jl:4-21 — SURVEY §8b puts it
at this boundary.)"""
import ctypes as C

import numpy as np
import pytest

from conftest import DEVICE_PAIRS


@pytest.mark.gpu
def test_hostile_arguments_return_errors(tfhe, orc, keys80):
    from tfhe_jl_amd import _lib as L
    lib = L.load()
    K = keys80
    eng = K.ck.engine(0)
    h = eng._h
    n1 = K.params.lwe_size + 1
    vp, NULL = C.c_void_p, None
    x = tfhe.encrypt(K.rng, K.sk, [True, False, True]).data
    y = tfhe.encrypt(K.rng, K.sk, [True, True, False]).data
    ops = np.zeros(3, np.uint8)
    out = np.zeros((3, n1), np.int32)
    p = lambda a: a.ctypes.data_as(vp)
    bad = []          # (what, status)

    def expect_error(what, rc):
        if rc == 0: bad.append(what)
        else: assert lib.tfhe_last_error(h) is not None

    # NULL context on every entry point that takes one
    for name, args in [("tfhe_load_bootstrap_key_i32", (NULL, NULL)), ("tfhe_load_keyswitch_key", (NULL, NULL)),
                       ("tfhe_gates_batch", (NULL, p(ops), p(x), p(y), NULL, p(out), 3)), ("tfhe_wires_alloc", (NULL, 4)),
                       ("tfhe_gates_batch_wait", (NULL, 0)), ("tfhe_set_option", (NULL, b"br_tiny", 1))]:
        assert getattr(lib, name)(*args) != 0, name + "(NULL ctx)"
    assert lib.tfhe_last_error(NULL) is not None
    # NULL / negative on the batch calls
    expect_error("gates_batch NULL opcodes", lib.tfhe_gates_batch(h, NULL, p(x), p(y), NULL, p(out), 3))
    expect_error("gates_batch NULL out", lib.tfhe_gates_batch(h, p(ops), p(x), p(y), NULL, NULL, 3))
    expect_error("gates_batch NULL in0", lib.tfhe_gates_batch(h, p(ops), NULL, p(y), NULL, p(out), 3))
    expect_error("gates_batch B < 0", lib.tfhe_gates_batch(h, p(ops), p(x), p(y), NULL, p(out), -1))
    expect_error("gates_batch B = 2^40", lib.tfhe_gates_batch(h, p(ops), p(x), p(y), NULL, p(out), 1 << 40))
    expect_error("gates_batch opcode 200", lib.tfhe_gates_batch(h, p(np.array([200, 0, 0], np.uint8)), p(x), p(y), NULL, p(out), 3))
    expect_error("gates_batch_dev NULL out", lib.tfhe_gates_batch_dev(h, p(ops), NULL, NULL, NULL, NULL, 3, NULL))
    expect_error("bootstrap_batch NULL in", lib.tfhe_bootstrap_batch(h, 1 << 29, NULL, p(out), 3, 1))
    expect_error("bootstrap_batch B < 0", lib.tfhe_bootstrap_batch(h, 1 << 29, p(x), p(out), -5, 1))
    expect_error("keyswitch_batch NULL", lib.tfhe_keyswitch_batch(h, NULL, NULL, 3))
    expect_error("gates_batch_wait bad ticket", lib.tfhe_gates_batch_wait(h, 12345))
    expect_error("gates_batch_wait negative ticket", lib.tfhe_gates_batch_wait(h, -1))
    # key loads
    expect_error("load_bootstrap_key NULL", lib.tfhe_load_bootstrap_key_i32(h, NULL))
    expect_error("load_bootstrap_key_c128 NULL", lib.tfhe_load_bootstrap_key_c128(h, NULL))
    expect_error("load_keyswitch_key NULL", lib.tfhe_load_keyswitch_key(h, NULL))
    expect_error("keygen NULL", lib.tfhe_keygen_cloud_key(h, NULL, NULL, C.c_double(1e-9), C.c_double(1e-5), NULL, NULL, NULL))
    # multi-key entry points on a single-key context
    expect_error("mk_load_bootstrap_key on single-key ctx", lib.tfhe_mk_load_bootstrap_key_i32(h, p(x), 2))
    expect_error("mk_load_keyswitch_key on single-key ctx", lib.tfhe_mk_load_keyswitch_key(h, p(x), 2))
    expect_error("mk_gate_nand on single-key ctx", lib.tfhe_mk_gate_nand_batch(h, p(x), p(y), p(out), 3))
    # wire table
    idx = np.array([0, 1, 2], np.int32)
    assert lib.tfhe_wires_alloc(h, 0) == 0          # (the session's engine may carry a table from an earlier test: 0 frees it)
    expect_error("gates_level without a table", lib.tfhe_gates_level(h, p(ops), p(idx), p(idx), NULL, p(idx), 3))
    expect_error("wires_upload without a table", lib.tfhe_wires_upload(h, 0, 3, p(x)))
    expect_error("wires_alloc negative", lib.tfhe_wires_alloc(h, -4))
    assert lib.tfhe_wires_alloc(h, 8) == 0
    expect_error("wires_upload past the end", lib.tfhe_wires_upload(h, 6, 3, p(x)))
    expect_error("wires_upload negative first", lib.tfhe_wires_upload(h, -1, 3, p(x)))
    expect_error("wires_upload NULL", lib.tfhe_wires_upload(h, 0, 3, NULL))
    expect_error("wires_download past the end", lib.tfhe_wires_download(h, 7, 2, p(out)))
    expect_error("wires_gather index 8", lib.tfhe_wires_gather(h, p(np.array([0, 8], np.int32)), 2, p(out)))
    expect_error("wires_gather negative index", lib.tfhe_wires_gather(h, p(np.array([-1, 0], np.int32)), 2, p(out)))
    expect_error("gates_level operand outside the table", lib.tfhe_gates_level(h, p(ops), p(np.array([0, 1, 9], np.int32)), p(idx), NULL, p(idx), 3))
    expect_error("gates_level output outside the table", lib.tfhe_gates_level(h, p(ops), p(idx), p(idx), NULL, p(np.array([3, 4, -2], np.int32)), 3))
    expect_error("gates_level NULL output indices", lib.tfhe_gates_level(h, p(ops), p(idx), p(idx), NULL, NULL, 3))
    expect_error("gates_level MUX without third operand", lib.tfhe_gates_level(h, p(np.full(3, 11, np.uint8)), p(idx), p(idx), NULL, p(np.array([3, 4, 5], np.int32)), 3))
    # options, diagnostics
    v = C.c_int64(0)
    f = C.c_float(0)
    d = C.c_double(0)
    expect_error("set_option unknown", lib.tfhe_set_option(h, b"no_such_option", 1))
    assert lib.tfhe_set_option(h, NULL, 1) == 0 and lib.tfhe_set_option(h, b"", 1) == 0      # no name, nothing to set: a no-op by design
    expect_error("get_option unknown", lib.tfhe_get_option(h, b"no_such_option", C.byref(v)))
    expect_error("get_option NULL out", lib.tfhe_get_option(h, b"br_tiny", NULL))
    expect_error("last_timing_ms which = 9", lib.tfhe_last_timing_ms(h, 9, C.byref(f)))
    expect_error("last_timing_ms NULL", lib.tfhe_last_timing_ms(h, 0, NULL))
    expect_error("timing_history NULL", lib.tfhe_timing_history_ms(h, 0, NULL, 4, NULL))
    expect_error("last_rounding_margin NULL", lib.tfhe_last_rounding_margin(h, NULL))
    # context creation
    hh = vp()
    P = L.TfheParams()
    lib.tfhe_ctx_params(h, C.byref(P))
    for field, val in [("N", 1000), ("N", 0), ("N", -1024), ("N", 16384), ("n", 0), ("n", -5), ("k", 0), ("bs_l", 0), ("bs_l", 5), ("ks_t", 0), ("ks_t", 40), ("parties", 0)]:
        Q = L.TfheParams()
        C.memmove(C.byref(Q), C.byref(P), C.sizeof(P))
        if not hasattr(Q, field): continue
        setattr(Q, field, val)
        rc = lib.tfhe_ctx_create(C.byref(Q), 0, C.byref(hh))
        if rc == 0:
            bad.append(f"ctx_create {field} = {val}")
            lib.tfhe_ctx_destroy(hh)
    assert lib.tfhe_ctx_create(NULL, 0, C.byref(hh)) != 0
    assert lib.tfhe_ctx_create(C.byref(P), 0, NULL) != 0
    assert lib.tfhe_ctx_create(C.byref(P), -3, C.byref(hh)) != 0
    assert lib.tfhe_ctx_create_multi(C.byref(P), NULL, 2, C.byref(hh)) != 0
    assert lib.tfhe_ctx_create_multi(C.byref(P), p(np.array([0], np.int32)), 0, C.byref(hh)) != 0
    assert lib.tfhe_ctx_create_multi(C.byref(P), p(np.array([0], np.int32)), -1, C.byref(hh)) != 0
    lib.tfhe_ctx_destroy(NULL)                # documented no-op
    lib.tfhe_host_free(NULL)
    assert not bad, f"accepted without an error: {bad}"
    # the context is still sound
    got = eng.gates(ops, x, y)
    assert np.array_equal(got, K.oracle.gates(ops, x, y, nthreads=4))
    assert list(tfhe.decrypt(K.sk, got)) == [False, True, True]


@pytest.mark.gpu
@pytest.mark.parametrize("devs", DEVICE_PAIRS)
def test_hostile_arguments_multi_key_and_multi_device(tfhe, orc, keys80, devs):
    """The same on a multi-key context (single-key entry points refused, NULL / negative arguments) and on a {0, 0} multi-device
    context (wire-table calls with indices outside the table, device-pointer calls)."""
    from tfhe_jl_amd import _lib as L
    lib = L.load()
    vp, NULL = C.c_void_p, None
    p = lambda a: a.ctypes.data_as(vp)
    bad = []

    def expect_error(what, rc):
        if rc == 0: bad.append(what)

    # multi-key, 2 parties, a small set
    P = tfhe.SchemeParameters(4, 0.012467, 1024, 1, 4, 7, 3.29e-10, 8, 2, 2.44e-5, 2)
    rng = np.random.default_rng(77)
    sks = [tfhe.SecretKey(rng, P) for _ in range(2)]
    shared = tfhe.SharedKey(rng, P)
    ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, sk, shared) for sk in sks])
    eng = ck.engine(0)
    h = eng._h
    w = 2 * 4 + 1
    x = tfhe.mk_encrypt(rng, sks, [True, False])
    y = tfhe.mk_encrypt(rng, sks, [True, True])
    out = np.zeros((2, w), np.int32)
    ops = np.zeros(2, np.uint8)
    expect_error("mk_gate_nand NULL in0", lib.tfhe_mk_gate_nand_batch(h, NULL, p(y), p(out), 2))
    expect_error("mk_gate_nand NULL out", lib.tfhe_mk_gate_nand_batch(h, p(x), p(y), NULL, 2))
    expect_error("mk_gate_nand B < 0", lib.tfhe_mk_gate_nand_batch(h, p(x), p(y), p(out), -2))
    expect_error("gates_batch on a multi-key ctx", lib.tfhe_gates_batch(h, p(ops), p(x), p(y), NULL, p(out), 2))
    expect_error("bootstrap_batch on a multi-key ctx", lib.tfhe_bootstrap_batch(h, 1 << 29, p(x), p(out), 2, 1))
    expect_error("load_keyswitch_key on a multi-key ctx", lib.tfhe_load_keyswitch_key(h, p(x)))
    expect_error("mk_load_bootstrap_key NULL", lib.tfhe_mk_load_bootstrap_key_i32(h, NULL, 2))
    expect_error("mk_load_bootstrap_key 0 parties", lib.tfhe_mk_load_bootstrap_key_i32(h, p(ck.bootstrap_key), 0))
    expect_error("mk_load_bootstrap_key 3 parties on max_parties = 2", lib.tfhe_mk_load_bootstrap_key_i32(h, p(ck.bootstrap_key), 3))
    expect_error("mk_load_keyswitch_key NULL", lib.tfhe_mk_load_keyswitch_key(h, NULL, 2))
    expect_error("mk_expand NULL parts", lib.tfhe_mk_expand_load_bootstrap_key(h, 2, NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL))
    o = orc.Oracle(4, 1024, 1, 4, 7, 8, 2, parties=2)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    assert np.array_equal(eng.mk_gate_nand(x, y), o.mk_gate_nand(x, y, nthreads=4))       # still sound
    # {0, 0} multi-device context
    K = keys80
    m = K.ck.engine(devs)
    hm = m._h
    n1 = K.params.lwe_size + 1
    a = tfhe.encrypt(K.rng, K.sk, [True, False, True]).data
    idx = np.array([0, 1, 2], np.int32)
    ops3 = np.zeros(3, np.uint8)
    o3 = np.zeros((3, n1), np.int32)
    assert lib.tfhe_wires_alloc(hm, 0) == 0
    expect_error("multi: gates_level without a table", lib.tfhe_gates_level(hm, p(ops3), p(idx), p(idx), NULL, p(idx), 3))
    assert lib.tfhe_wires_alloc(hm, 8) == 0
    expect_error("multi: wires_upload past the end", lib.tfhe_wires_upload(hm, 7, 3, p(a)))
    expect_error("multi: wires_gather index 99", lib.tfhe_wires_gather(hm, p(np.array([0, 99], np.int32)), 2, p(o3)))
    expect_error("multi: gates_level operand outside", lib.tfhe_gates_level(hm, p(ops3), p(np.array([0, 1, 8], np.int32)), p(idx), NULL, p(np.array([3, 4, 5], np.int32)), 3))
    expect_error("multi: gates_level NULL opcodes", lib.tfhe_gates_level(hm, NULL, p(idx), p(idx), NULL, p(np.array([3, 4, 5], np.int32)), 3))
    expect_error("multi: gates_batch_dev (device pointers belong to one device)", lib.tfhe_gates_batch_dev(hm, p(ops3), p(a), p(a), NULL, p(o3), 3, NULL))
    expect_error("multi: gates_batch NULL out", lib.tfhe_gates_batch(hm, p(ops3), p(a), p(a), NULL, NULL, 3))
    expect_error("multi: level_exchange = 7", lib.tfhe_set_option(hm, b"level_exchange", 7))
    assert not bad, f"accepted without an error: {bad}"
    b = tfhe.encrypt(K.rng, K.sk, [True, True, False]).data
    assert np.array_equal(m.gates(ops3, a, b), K.oracle.gates(ops3, a, b, nthreads=4))
