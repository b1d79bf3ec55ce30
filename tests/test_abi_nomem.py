"""No C++ exception crosses the C ABI (include/tfhe_mi355x.h, "no exceptions cross the boundary"; SURVEY §8b "Error conventions"):
an allocation failure inside the library comes back as TFHE_ERR_NOMEM with a message, and the context goes on working.

The failure is injected: tfhe_set_option(ctx or NULL, "debug_fail_alloc_after", n) makes the n-th allocation checkpoint from
then on throw std::bad_alloc (csrc/engine.hpp: alloc_checkpoint — at the start of every entry point and in front of the host
allocations that grow with the batch).  Under a Julia `ccall` or Python `ctypes` an exception that escaped would end the process:
that these tests finish at all is the first thing they show."""
import ctypes as C
import threading

import numpy as np
import pytest

from conftest import DEVICE_PAIRS, device_count

NOMEM = 6


def _lib():
    from tfhe_jl_amd import _lib as L
    return L, L.load()


def test_nomem_status_without_a_device():
    """Entry points that need no context, on any machine: the injected std::bad_alloc is caught, the status is TFHE_ERR_NOMEM,
    tfhe_last_error(NULL) says so, and the next call behaves as if nothing had happened."""
    L, lib = _lib()
    vp = C.c_void_p
    P = L.TfheParams(16, 1024, 1, 2, 10, 8, 2, 1)
    bounds = np.zeros(3, np.int64)
    ids = np.array([0, 0], np.int32)
    h, mem, v = vp(), vp(), C.c_int64(-1)

    def armed(n=1):
        assert lib.tfhe_set_option(None, b"debug_fail_alloc_after", n) == 0
        assert lib.tfhe_get_option(None, b"debug_fail_alloc_after", C.byref(v)) == 0 and v.value == n

    calls = {
        "tfhe_shard_bounds": lambda: lib.tfhe_shard_bounds(None, 10, 2, bounds.ctypes.data_as(vp)),
        "tfhe_ctx_create": lambda: lib.tfhe_ctx_create(C.byref(P), 0, C.byref(h)),
        "tfhe_ctx_create_multi": lambda: lib.tfhe_ctx_create_multi(C.byref(P), ids.ctypes.data_as(vp), 2, C.byref(h)),
        "tfhe_host_alloc": lambda: lib.tfhe_host_alloc(4096, C.byref(mem)),
    }
    for name, call in calls.items():
        armed()
        assert call() == NOMEM, name
        msg = lib.tfhe_last_error(None).decode()
        assert name in msg and "memory" in msg, msg
        assert lib.tfhe_get_option(None, b"debug_fail_alloc_after", C.byref(v)) == 0 and v.value == 0      # consumed: one failure, then off
    # ... and unarmed the same calls do what they always did
    assert lib.tfhe_shard_bounds(None, 10, 2, bounds.ctypes.data_as(vp)) == 0 and list(bounds) == [0, 5, 10]
    rc = lib.tfhe_ctx_create(C.byref(P), 0, C.byref(h))
    assert rc in (0, 4)                     # a context, or "no HIP device" on the CPU-only container — never NOMEM, never a crash
    if rc == 0:
        lib.tfhe_ctx_destroy(h)
    assert lib.tfhe_set_option(None, b"debug_fail_alloc_after", -1) != 0
    assert lib.tfhe_set_option(None, b"br_tiny", 1) != 0                      # every other option still needs a context


@pytest.mark.gpu
def test_allocation_failure_in_every_entry_point_leaves_the_context_sound(tfhe, orc, keys80):
    """One-device and {0, 0} multi-device contexts: for n = 1, 2, ... the n-th checkpoint of a call fails until the call gets
    through; every failed attempt returns TFHE_ERR_NOMEM with a message, and after each one the SAME context computes the oracle's
    words.  The walk covers the checkpoints inside run_gates / validate_level / pull_wires / the shard tables and those hit on the
    fan-out's worker threads."""
    L, lib = _lib()
    K = keys80
    vp = C.c_void_p
    p = lambda a: a.ctypes.data_as(vp)
    n1 = K.params.lwe_size + 1
    B = 6
    bits = K.rng.integers(0, 2, (3, B)).astype(bool)
    x, y, z = (tfhe.encrypt(K.rng, K.sk, b).data for b in bits)
    ops = np.array([0, 11, 3, 5, 11, 2], np.uint8)              # NAND, MUX, XOR, NOT, MUX, AND
    want = K.oracle.gates(ops, x, y, z, nthreads=4)
    idx = np.arange(B, dtype=np.int32)

    def walk(h, what, call, check):
        failures = 0
        for n in range(1, 200):
            assert lib.tfhe_set_option(None, b"debug_fail_alloc_after", n) == 0
            rc = call()
            lib.tfhe_set_option(None, b"debug_fail_alloc_after", 0)
            if rc == 0:
                check()
                return failures
            assert rc == NOMEM, (what, n, rc, lib.tfhe_last_error(h))
            assert b"memory" in lib.tfhe_last_error(h), (what, n, lib.tfhe_last_error(h))
            failures += 1
        raise AssertionError(f"{what}: still failing after 200 checkpoints")

    for devices in (None, [0, 0], [0, 1])[:3 if device_count() >= 2 else 2]:
        eng = K.ck.engine(0) if devices is None else K.ck.engine(devices)
        h = eng._h
        out = np.zeros((B, n1), np.int32)
        counts = {}

        def sound():
            got = eng.gates(ops, x, y, z)
            assert np.array_equal(got, want)

        def check_out():
            assert np.array_equal(out, want)
            out[:] = 0

        counts["gates_batch"] = walk(h, "gates_batch", lambda: lib.tfhe_gates_batch(h, p(ops), p(x), p(y), p(z), p(out), B), check_out)
        sound()
        ticket = C.c_int32(-1)

        def submit_and_wait():
            rc = lib.tfhe_gates_batch_submit(h, p(ops), p(x), p(y), p(z), p(out), B, C.byref(ticket))
            return rc or lib.tfhe_gates_batch_wait(h, ticket.value)
        counts["submit"] = walk(h, "gates_batch_submit", submit_and_wait, check_out)
        sound()
        ob = np.zeros((B, n1), np.int32)
        wb = K.oracle.bootstrap(2**29, x, nthreads=4)
        counts["bootstrap"] = walk(h, "bootstrap_batch", lambda: lib.tfhe_bootstrap_batch(h, 2**29, p(x), p(ob), B, 1), lambda: np.testing.assert_array_equal(ob, wb))
        sound()
        # wire table: alloc, upload, one level (on the multi-device context: sharded, rows pulled across), gather
        lib.tfhe_set_option(h, b"level_split_min", 2)
        counts["wires_alloc"] = walk(h, "wires_alloc", lambda: lib.tfhe_wires_alloc(h, 4 * B), lambda: None)
        rows = np.concatenate([x, y, z])
        counts["wires_upload"] = walk(h, "wires_upload", lambda: lib.tfhe_wires_upload(h, 0, 3 * B, p(rows)), lambda: None)
        o_idx = (3 * B + idx).astype(np.int32)
        lvl = lambda: lib.tfhe_gates_level(h, p(ops), p(idx), p((B + idx).astype(np.int32)), p((2 * B + idx).astype(np.int32)), p(o_idx), B)
        got = np.zeros((B, n1), np.int32)
        counts["gates_level"] = walk(h, "gates_level", lvl, lambda: None)
        counts["wires_gather"] = walk(h, "wires_gather", lambda: lib.tfhe_wires_gather(h, p(o_idx), B, p(got)), lambda: np.testing.assert_array_equal(got, want))
        # a second level that reads what the first wrote (multi-device: rows travel between the replicas), after failures in between
        ops2 = np.full(B, 14, np.uint8)             # COPY
        lvl2 = lambda: lib.tfhe_gates_level(h, p(ops2), p(o_idx), None, None, p(idx), B)
        counts["gates_level 2"] = walk(h, "gates_level (reads the first level's rows)", lvl2, lambda: None)
        dl = np.zeros((B, n1), np.int32)
        counts["wires_download"] = walk(h, "wires_download", lambda: lib.tfhe_wires_download(h, 0, B, p(dl)), lambda: np.testing.assert_array_equal(dl, want))
        sound()
        assert all(c >= 1 for c in counts.values()), counts            # every walk saw at least the entry checkpoint fail
        if devices is not None:
            assert counts["gates_level"] >= 3 and counts["gates_batch"] >= 3, counts      # ... and the deeper ones on the multi-device paths
        print(f"devices {devices}: injected failures per entry point {counts}")
        lib.tfhe_set_option(h, b"level_split_min", 4096)
        assert lib.tfhe_wires_alloc(h, 0) == 0          # (the engine is the session's: leave it as it was found)


@pytest.mark.gpu
def test_key_loaders_survive_an_allocation_failure(tfhe, orc):
    """tfhe_keygen_cloud_key (staging vectors, secret material zeroed on the way out) and the multi-key expansion (the host-side
    digit table) under the same walk; the keys loaded by the attempt that got through are the ones a clean context generates."""
    L, lib = _lib()
    vp = C.c_void_p
    p = lambda a: a.ctypes.data_as(vp)
    P = tfhe.SchemeParameters(12, 1 / 2**15, 1024, 1, 2, 10, 9e-9, 8, 2, 1 / 2**15, 1)
    rng = np.random.default_rng(5)
    sk = tfhe.SecretKey(rng, P)
    tl = rng.integers(0, 2, 1024).astype(np.int32)
    seed = np.arange(1, 7, dtype=np.uint32)
    ref = tfhe._lib.Engine(P)
    bk_ref, ks_ref = ref.keygen_cloud_key(sk.key.key, tl, P.bs_noise_stddev, P.ks_noise_stddev, seed)
    eng = tfhe._lib.Engine(P, devices=[0, 0])
    h = eng._h
    bk = np.zeros_like(bk_ref); ks = np.zeros_like(ks_ref)
    lwe = np.ascontiguousarray(sk.key.key, np.int32)
    fails = 0
    for n in range(1, 100):
        lib.tfhe_set_option(None, b"debug_fail_alloc_after", n)
        rc = lib.tfhe_keygen_cloud_key(h, p(lwe), p(tl), C.c_double(P.bs_noise_stddev), C.c_double(P.ks_noise_stddev), p(seed), p(bk), p(ks))
        lib.tfhe_set_option(None, b"debug_fail_alloc_after", 0)
        if rc == 0: break
        assert rc == NOMEM, (n, rc, lib.tfhe_last_error(h))
        fails += 1
    assert rc == 0 and fails >= 1
    assert np.array_equal(bk, bk_ref) and np.array_equal(ks, ks_ref)
    x = rng.integers(-2**31, 2**31, size=(3, P.lwe_size + 1), dtype=np.int64).astype(np.int32)
    assert np.array_equal(eng.bootstrap(2**29, x), ref.bootstrap(2**29, x))
    eng.close(); ref.close()


@pytest.mark.gpu
def test_wait_and_synchronize_from_another_thread_during_a_running_call(tfhe, orc, keys80):
    """tfhe_gates_batch_wait and tfhe_ctx_synchronize are callable from any thread while another thread is inside a call on the
    same context (ABI v7) — what a finalizer needs before it frees the page-locked buffers of a submitted batch.  Until v6 the
    wait returned TFHE_ERR_STATE at once in that situation (and the Julia shim then freed the buffers under a live DMA)."""
    L, lib = _lib()
    K = keys80
    eng = K.ck.engine(0)
    h = eng._h
    n1 = K.params.lwe_size + 1
    B = 2048
    rng = np.random.default_rng(11)
    a = tfhe.encrypt(rng, K.sk, rng.integers(0, 2, B).astype(bool)).data
    b = tfhe.encrypt(rng, K.sk, rng.integers(0, 2, B).astype(bool)).data
    ops = np.zeros(B, np.uint8)
    want = K.oracle.gates(ops[:64], a[:64], b[:64], nthreads=8)
    pa, pb, po = L.pinned_empty((B, n1)), L.pinned_empty((B, n1)), L.pinned_empty((B, n1))
    pa[:] = a; pb[:] = b; po[:] = 0
    # 1. a submitted batch, then the owner thread goes into a long blocking call; a second thread waits for the ticket meanwhile
    ticket, _ = eng.gates_submit(ops, pa, pb, out=po)
    seen = {}
    started = threading.Event()

    def other_thread():
        started.wait()
        rcs = []
        for _ in range(50):
            rcs.append(lib.tfhe_gates_batch_wait(h, ticket))
            rcs.append(lib.tfhe_ctx_synchronize(h))
        seen["rcs"] = rcs
        seen["rows_after_first_wait_equal"] = bool(np.array_equal(po[:64], want))       # the wait returned: the batch's DMA is over
    t = threading.Thread(target=other_thread)
    t.start()
    outs = []
    for i in range(6):
        if i == 1: started.set()
        outs.append(eng.gates(ops, a, b))            # blocking calls by the owner of the context
    t.join()
    assert all(rc == 0 for rc in seen["rcs"]), seen["rcs"]
    assert seen["rows_after_first_wait_equal"]
    assert all(np.array_equal(o, outs[0]) for o in outs) and np.array_equal(outs[0][:64], want) and np.array_equal(po, outs[0])
    eng.gates_wait(ticket)                           # the owner's own wait afterwards: the slot is released as before
    # 2. the other direction: calls that DO need the context still refuse to overlap
    assert np.array_equal(eng.gates(ops[:64], a[:64], b[:64]), want)


@pytest.mark.gpu
def test_mk_keyswitch_key_for_fewer_parties_is_refused(tfhe):
    """a bootstrapping key loaded for P parties and a keyswitch key for fewer made the keyswitch loop read past
    the key.  Now the gate call says TFHE_ERR_STATE."""
    L, lib = _lib()
    P = tfhe.SchemeParameters(4, 0.012467, 1024, 1, 4, 7, 3.29e-10, 8, 2, 2.44e-5, 3)
    rng = np.random.default_rng(3)
    sks = [tfhe.SecretKey(rng, P) for _ in range(3)]
    shared = tfhe.SharedKey(rng, P)
    parts = [tfhe.CloudKeyPart(rng, sk, shared) for sk in sks]
    ck3 = tfhe.MKCloudKey(parts)
    ck2 = tfhe.MKCloudKey(parts[:2])
    eng = tfhe._lib.Engine(P)
    eng.mk_load_bootstrap_key(ck3.bootstrap_key, 3)
    eng.mk_load_keyswitch_key(ck2.keyswitch_key, 2)
    x = tfhe.mk_encrypt(rng, sks, [True, False])
    with pytest.raises(L.EngineError) as e:
        eng.mk_gate_nand(x, x)
    assert e.value.code == 5 and "parties" in str(e.value)
    eng.mk_load_keyswitch_key(ck3.keyswitch_key, 3)
    out = eng.mk_gate_nand(x, x)
    assert list(tfhe.mk_decrypt(sks, out)) == [False, True]
    eng.close()
