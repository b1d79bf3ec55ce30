"""GPU parity tests: the HIP path through the C ABI vs the oracle on identical keys and inputs.
Bit-exact on every Int32 word.  Run with -m gpu on an MI355X."""
import itertools

import numpy as np
import pytest

from test_oracle import GATES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng80(tfhe, keys80):
    return keys80.ck.engine(0)


@pytest.fixture(scope="module")
def eng128(tfhe, keys128):
    return keys128.ck.engine(0)


def test_native_library_is_the_one_loaded(tfhe, eng80):
    import os
    assert os.path.exists(tfhe.LIB_PATH)
    maps = open("/proc/self/maps").read()
    assert "libtfhe_mi355x.so" in maps


@pytest.mark.parametrize("name,nargs,ref", GATES, ids=[g[0] for g in GATES])
def test_gate_parity_and_truth_table_80(tfhe, orc, keys80, eng80, name, nargs, ref):
    K = keys80
    combos = list(itertools.product((False, True), repeat=nargs)) * 2
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(nargs)]
    ops = np.full(len(combos), tfhe.OPCODES[name], np.uint8)
    got = eng80.gates(ops, *ins)
    want = K.oracle.gates(ops, *ins, nthreads=8)
    assert np.array_equal(got, want)
    assert list(tfhe.decrypt(K.sk, got)) == [bool(ref(*c)) for c in combos]


def test_reference_api_mirror(tfhe, keys80, eng80):
    """gate_*(ck, x, y) with the reference's names: scalar, vector, constant, not."""
    K = keys80
    t, f = tfhe.encrypt(K.rng, K.sk, True), tfhe.encrypt(K.rng, K.sk, False)
    assert tfhe.decrypt(K.sk, tfhe.gate_nand(K.ck, t, t)) is False
    assert tfhe.decrypt(K.sk, tfhe.gate_mux(K.ck, f, t, f)) is False
    assert tfhe.decrypt(K.sk, tfhe.gate_not(K.ck, f)) is True
    assert tfhe.decrypt(K.sk, tfhe.gate_constant(K.ck, True)) is True
    xs = tfhe.encrypt(K.rng, K.sk, [True, False, True, False])
    ys = tfhe.encrypt(K.rng, K.sk, [True, True, False, False])
    assert list(tfhe.decrypt(K.sk, tfhe.gate_xor(K.ck, xs, ys))) == [False, True, True, False]


def test_mixed_gate_stream_parity(tfhe, orc, keys80, eng80):
    """BASELINE config 3 in miniature: i.i.d. opcodes over {NAND, AND, OR, XOR, MUX} + the trivial ones."""
    K = keys80
    rng = np.random.default_rng(789)
    names = ["NAND", "AND", "OR", "XOR", "MUX", "NOT", "CONST1", "COPY", "XNOR", "ORYN"]
    B = 96
    ops = np.array([tfhe.OPCODES[names[i]] for i in rng.integers(0, len(names), B)], np.uint8)
    ins = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data for _ in range(3)]
    got = eng80.gates(ops, *ins)
    want = K.oracle.gates(ops, *ins, nthreads=8)
    assert np.array_equal(got, want)
    n_mux = int(np.sum(ops == tfhe.OPCODES["MUX"]))
    n_triv = int(np.sum(np.isin(ops, [tfhe.OPCODES[x] for x in ("NOT", "CONST1", "COPY")])))
    assert eng80.last_rotation_count() == B - n_triv + n_mux


def test_bootstrap_and_keyswitch_entry_points(tfhe, orc, keys80, eng80):
    K = keys80
    rng = np.random.default_rng(42)
    x = rng.integers(-2**31, 2**31, size=(16, 501), dtype=np.int64).astype(np.int32)   # arbitrary words
    x[0, :] = 0
    x[1, :7] = [2**31 - 1, -2**31, 2**20, 2**20 - 1, -2**20, -2**20 - 1, 1]
    mu = 2**29
    ext = eng80.bootstrap(mu, x, with_keyswitch=False)
    assert np.array_equal(ext, K.oracle.bootstrap(mu, x, with_keyswitch=False, nthreads=8))
    assert np.array_equal(eng80.keyswitch(ext), K.oracle.keyswitch(ext))
    assert np.array_equal(eng80.bootstrap(mu, x, with_keyswitch=True), K.oracle.bootstrap(mu, x, nthreads=8))
    # a different mu (bootstrap's argument, bootstrap.jl:92)
    assert np.array_equal(eng80.bootstrap(-12345, x[:2]), K.oracle.bootstrap(-12345, x[:2]))
    # keyswitch on arbitrary (non-bootstrapped) words, incl. all-zero digits
    y = rng.integers(-2**31, 2**31, size=(8, 1025), dtype=np.int64).astype(np.int32)
    y[0, :] = 0
    y[1, :-1] = -2**15    # aibar = 0: every digit zero
    assert np.array_equal(eng80.keyswitch(y), K.oracle.keyswitch(y))


def test_gate_parity_128(tfhe, orc, keys128, eng128):
    K = keys128
    combos = list(itertools.product((False, True), repeat=3)) * 2
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(3)]
    for name in ("NAND", "MUX", "XOR"):
        ops = np.full(len(combos), tfhe.OPCODES[name], np.uint8)
        got = eng128.gates(ops, *ins)
        assert np.array_equal(got, K.oracle.gates(ops, *ins, nthreads=8))
    got = eng128.gates(np.full(len(combos), tfhe.OPCODES["NAND"], np.uint8), *ins)
    assert list(tfhe.decrypt(K.sk, got)) == [not (c[0] and c[1]) for c in combos]


def test_spectra_key_load_equals_int32_load(tfhe, orc, keys80, eng80):
    """tfhe_load_bootstrap_key_c128 (the reference's stored form, bootstrap.jl:12-14) gives the same engine."""
    K = keys80
    e2 = tfhe.Engine(K.params, 0)
    e2.load_bootstrap_key_spectra(K.oracle.bk_spectra())
    e2.load_keyswitch_key(K.ck.keyswitch_key)
    x = tfhe.encrypt(K.rng, K.sk, [True, False, True, False]).data
    y = tfhe.encrypt(K.rng, K.sk, [True, True, False, False]).data
    ops = np.full(4, tfhe.OPCODES["NAND"], np.uint8)
    assert np.array_equal(e2.gates(ops, x, y), eng80.gates(ops, x, y))
    e2.close()


def test_edge_cases_and_errors(tfhe, keys80, eng80):
    K = keys80
    empty = np.zeros((0, 501), np.int32)
    assert eng80.gates(np.zeros(0, np.uint8), empty, empty).shape == (0, 501)
    x = tfhe.encrypt(K.rng, K.sk, [True]).data
    with pytest.raises(tfhe.EngineError):
        eng80.gates(np.array([99], np.uint8), x, x)          # bad opcode
    with pytest.raises(tfhe.EngineError):
        eng80.gates(np.array([tfhe.OPCODES["MUX"]], np.uint8), x, x, None)   # MUX needs a third operand
    e = tfhe.Engine(K.params, 0)
    with pytest.raises(tfhe.EngineError) as ei:
        e.gates(np.array([0], np.uint8), x, x)               # keys not loaded
    assert ei.value.code == 3
    e.close()
    with pytest.raises(tfhe.EngineError):
        tfhe.Engine(K.params, 99)                            # no such device


@pytest.mark.parametrize("B,kernel", [(4096, "blind_rotate_kernel_v3<2,8,tw2reg,rw4>"), (1024, "blind_rotate_kernel_w2<2,rw2>"), (700, "blind_rotate_kernel_w2<2>"), (400, "blind_rotate_kernel_w2<2,rw2>")])
def test_full_batch_properties(tfhe, orc, keys80, eng80, B, kernel):
    """BASELINE config 2 size (4096: two rounds of the one-wave kernel, four rotations per workgroup in lockstep) and the chip-filling sizes of the two-wave kernel
    (1024: two pairs of rotations on every CU, the waves of a rotation swapping LDS buffers every step; 700: partly filled,
    single rotations; 400: at most one pair per CU):
    every output decrypts to NAND; at 4096 ALL outputs are bit-equal to the oracle (the oracle on every
    host thread: seconds), at the other sizes every row as well (they are smaller); the batch is deterministic and independent of
    batch position."""
    K = keys80
    rng = np.random.default_rng(456)
    bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.encrypt(K.rng, K.sk, bx).data, tfhe.encrypt(K.rng, K.sk, by).data
    ops = np.zeros(B, np.uint8)
    got = eng80.gates(ops, x, y)
    assert np.array_equal(tfhe.decrypt(K.sk, got), ~(bx & by))
    assert np.array_equal(got, K.oracle.gates(ops, x, y, nthreads=orc.max_threads()))
    perm = rng.permutation(B)
    got2 = eng80.gates(ops, x[perm], y[perm])
    assert np.array_equal(got2, got[perm])
    assert eng80.last_rotation_count() == B and eng80.last_kernel_name() == kernel
    assert eng80.last_timing_ms(0) > 0 and eng80.last_timing_ms(1) > 0
    for _ in range(3):                                   # same launch again: no run-to-run difference (a race would show here)
        assert np.array_equal(eng80.gates(ops, x, y), got)
    for pct in (0, 50, 100):                             # the issue-priority schedule changes timing only
        eng80.set_option("br_prio_pct", pct)
        assert np.array_equal(eng80.gates(ops, x, y), got), pct
    eng80.set_option("br_prio_pct", 90)


def test_tutorial_encrypted_minimum(tfhe, keys80, eng80):
    """examples/tutorial.jl:42-78 end to end: the 16-bit encrypted minimum of 2017 and 42 decrypts to 42
    (16 XNOR + 32 MUX = 80 blind rotations, 48 keyswitches), through the reference-named gate API."""
    K = keys80
    a_bits = [(2017 >> i) & 1 == 1 for i in range(16)]          # tutorial.jl:24-26
    b_bits = [(42 >> i) & 1 == 1 for i in range(16)]            # :29-31
    a = [tfhe.encrypt(K.rng, K.sk, bit) for bit in a_bits]
    b = [tfhe.encrypt(K.rng, K.sk, bit) for bit in b_bits]
    carry = tfhe.gate_constant(K.ck, False)                     # :52
    for i in range(16):                                         # :54-56, compare_bit :42-45
        tmp = tfhe.gate_xnor(K.ck, a[i], b[i])
        carry = tfhe.gate_mux(K.ck, tmp, carry, a[i])
    sel = tfhe.LweSampleArray.from_samples([carry] * 16)
    res = tfhe.gate_mux(K.ck, sel, tfhe.LweSampleArray.from_samples(b), tfhe.LweSampleArray.from_samples(a))   # :60 as one batch
    bits = tfhe.decrypt(K.sk, res)
    assert sum(int(v) << i for i, v in enumerate(bits)) == 42   # :72-77 "Answer: 42"


def test_gate_parity_mask_size_2(tfhe, orc):
    """tlwe_mask_size = 2 (api.jl:30 keyword): 3-polynomial accumulator kernel, 2N-word keyswitch."""
    from conftest import KeySet
    K = KeySet(tfhe, orc, tfhe.tfhe_parameters_80(tlwe_mask_size=2), seed=77)
    eng = K.ck.engine(0)
    combos = list(itertools.product((False, True), repeat=3))
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(3)]
    for name in ("NAND", "MUX", "XNOR"):
        ops = np.full(8, tfhe.OPCODES[name], np.uint8)
        got = eng.gates(ops, *ins)
        assert np.array_equal(got, K.oracle.gates(ops, *ins, nthreads=8))
    got = eng.gates(np.full(8, tfhe.OPCODES["MUX"], np.uint8), *ins)
    assert list(tfhe.decrypt(K.sk, got)) == [bool(y if x else z) for x, y, z in combos]
    x = K.rng.integers(-2**31, 2**31, size=(4, 501), dtype=np.int64).astype(np.int32)
    ext = eng.bootstrap(2**29, x, with_keyswitch=False)
    assert ext.shape == (4, 2049) and np.array_equal(ext, K.oracle.bootstrap(2**29, x, with_keyswitch=False))
    assert np.array_equal(eng.keyswitch(ext), K.oracle.keyswitch(ext))
    K.ck.close()


def test_gate_parity_synthetic_n2048(tfhe, orc):
    """BASELINE config 4b (synthetic N = 2048, l = 3, beta = 7): two-wave blind-rotate kernel."""
    from conftest import KeySet
    from test_oracle import synthetic_2048
    K = KeySet(tfhe, orc, synthetic_2048(tfhe), seed=2048)
    eng = K.ck.engine(0)
    combos = list(itertools.product((False, True), repeat=3))
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(3)]
    for name in ("NAND", "MUX"):
        ops = np.full(8, tfhe.OPCODES[name], np.uint8)
        got = eng.gates(ops, *ins)
        assert np.array_equal(got, K.oracle.gates(ops, *ins, nthreads=8))
    got = eng.gates(np.full(8, tfhe.OPCODES["MUX"], np.uint8), *ins)
    assert list(tfhe.decrypt(K.sk, got)) == [bool(y if x else z) for x, y, z in combos]
    x = K.rng.integers(-2**31, 2**31, size=(3, 631), dtype=np.int64).astype(np.int32)
    x[0, :3] = [2**31 - 1, -2**31, 2**19]
    ext = eng.bootstrap(2**29, x, with_keyswitch=False)
    assert ext.shape == (3, 2049) and np.array_equal(ext, K.oracle.bootstrap(2**29, x, with_keyswitch=False, nthreads=3))
    # the reference's stored spectra load to the same engine state
    e2 = tfhe.Engine(K.params, 0)
    e2.load_bootstrap_key_spectra(K.oracle.bk_spectra())
    e2.load_keyswitch_key(K.ck.keyswitch_key)
    assert np.array_equal(e2.bootstrap(2**29, x, with_keyswitch=False), ext)
    e2.close()
    K.ck.close()


@pytest.mark.parametrize("B", [2, 31, 33, 65, 255, 257, 513, 1027, 1300, 2600])
def test_ragged_batch_sizes(tfhe, orc, keys80, eng80, B):
    """Batch sizes straddling every tile edge (MFMA keyswitch tiles of 32/64/256 samples, the 1024-rotation switch
    between the two-wave and one-wave blind-rotate kernels, a mixed batch whose rotation count leaves a part-filled last
    round (2600 gates: the split launch) — MUX counts twice, NOT not at all): all bit-equal to the oracle."""
    K = keys80
    rng = np.random.default_rng(B)
    names = ["NAND", "XOR", "MUX", "NOT", "ANDYN"]
    ops = np.array([tfhe.OPCODES[names[i]] for i in rng.integers(0, len(names), B)], np.uint8)
    ins = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data for _ in range(3)]
    got = eng80.gates(ops, *ins)
    assert np.array_equal(got, K.oracle.gates(ops, *ins, nthreads=16))


@pytest.mark.parametrize("B", [2560, 3072, 5000])
def test_split_dispatch_of_part_filled_rounds(tfhe, orc, keys80, eng80, B):
    """Batches above what the chip holds whose last round is at most 1024 rotations: the whole rounds go to the one-wave
    kernel, the tail to the two-wave kernel in a second launch (launch_blind_rotate, round 4).  Rows around the seam and at
    both ends equal the oracle; the whole batch equals the single-launch result (option br_split 0) word for word."""
    K = keys80
    rng = np.random.default_rng(B)
    bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.encrypt(K.rng, K.sk, bx).data, tfhe.encrypt(K.rng, K.sk, by).data
    ops = np.zeros(B, np.uint8)
    eng80.set_option("pipeline_min", -1)           # one call = one batch (no two-stream halves)
    try:
        got = eng80.gates(ops, x, y)
        name = eng80.last_kernel_name()
        head = B - B % 2048
        assert name.startswith("blind_rotate_kernel_v3<2,8,tw2reg,rw4> + blind_rotate_kernel_w2<2"), name
        idx = sorted({0, 1, head - 2, head - 1, head, head + 1, B - 2, B - 1} | set(int(v) for v in rng.choice(B, 24, replace=False)))
        assert np.array_equal(got[idx], K.oracle.gates(ops[idx], x[idx], y[idx], nthreads=16))
        assert np.array_equal(tfhe.decrypt(K.sk, got), ~(bx & by))
        eng80.set_option("br_split", 0)
        one = eng80.gates(ops, x, y)
        assert eng80.last_kernel_name() == "blind_rotate_kernel_v3<2,8,tw2reg,rw4>"
        assert np.array_equal(one, got)
    finally:
        eng80.set_option("br_split", 1)
        eng80.set_option("pipeline_min", 4096)


def test_ragged_batch_mask_size_2(tfhe, orc):
    """k = 2 (tlwe_mask_size keyword, api.jl:30): a batch that is not a multiple of the 1792 rotations a round of the lockstep
    groups holds — 2000 mixed gates, rows from every round and the seams against the oracle."""
    from conftest import KeySet
    K = KeySet(tfhe, orc, tfhe.SchemeParameters(24, 1 / 2**15, 1024, 2, 2, 10, 9e-9, 8, 2, 1 / 2**15, 1), seed=777)
    eng = K.ck.engine(0)
    B = 2000
    rng = np.random.default_rng(2)
    names = ["NAND", "XOR", "MUX", "NOT", "ANDYN"]
    ops = np.array([tfhe.OPCODES[names[i]] for i in rng.integers(0, len(names), B)], np.uint8)
    ins = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data for _ in range(3)]
    got = eng.gates(ops, *ins)
    # (about 2000 rotations = 8 per CU: one full round of seven per CU on the lockstep groups, the rest on the three-wave kernel)
    assert eng.last_kernel_name() == "blind_rotate_kernel_k2<2,rw7> + blind_rotate_kernel_k2w3<2>", eng.last_kernel_name()
    idx = sorted({0, 1, 255, 256, 1791, 1792, 1793, B - 1} | set(int(v) for v in rng.choice(B, 40, replace=False)))
    assert np.array_equal(got[idx], K.oracle.gates(ops[idx], *[a[idx] for a in ins], nthreads=16))
    K.ck.close()


def test_mask_size_2_round_partition(tfhe, orc):
    """k = 2, 4096 rotations = 16 per CU: two rounds of seven per CU on the lockstep groups and the last 512 rotations on the
    three-waves-per-rotation kernel (k2_partition; round 4 dealt 6 + 6 + 4, all on the one-wave kernel: option k2_w3 = 0) — same
    words as the single launch, rows from every segment and the seams equal the oracle."""
    from conftest import KeySet
    K = KeySet(tfhe, orc, tfhe.SchemeParameters(16, 1 / 2**15, 1024, 2, 2, 10, 9e-9, 8, 2, 1 / 2**15, 1), seed=778)
    eng = K.ck.engine(0)
    R = 4096
    rng = np.random.default_rng(3)
    x = rng.integers(-2**31, 2**31, size=(R, 17), dtype=np.int64).astype(np.int32)
    got = eng.bootstrap(2**29, x, with_keyswitch=False)
    assert eng.last_kernel_name() == "blind_rotate_kernel_k2<2,rw7> + blind_rotate_kernel_k2w3<2>", eng.last_kernel_name()
    idx = sorted({0, 1, 3071, 3072, 3073, 3583, 3584, 3585, R - 1} | set(int(v) for v in rng.choice(R, 26, replace=False)))
    assert np.array_equal(got[idx], K.oracle.bootstrap(2**29, x[idx], with_keyswitch=False, nthreads=16))
    eng.set_option("k2_w3", 0)                # round 4's partition: 6 + 6 + 4 per CU in two launches of the one-wave kernel
    assert np.array_equal(eng.bootstrap(2**29, x, with_keyswitch=False), got)
    assert eng.last_kernel_name() == "blind_rotate_kernel_k2<2,rw7>"
    eng.set_option("br_split", 0)
    assert np.array_equal(eng.bootstrap(2**29, x, with_keyswitch=False), got)
    K.ck.close()


def test_streamed_batches_equal_synchronous(tfhe, orc, keys80, eng80):
    """tfhe_gates_batch_submit / _wait: six batches of different sizes and opcode mixes streamed two at a time from
    page-locked buffers — every result bit-equal to the blocking call's; a third submit displaces (waits for) the oldest
    batch; waiting twice, or for a displaced ticket, is harmless; a ticket nobody issued is an error."""
    K = keys80
    rng = np.random.default_rng(77)
    names = ["NAND", "XOR", "MUX", "NOT", "ANDYN", "CONST1"]
    jobs = []
    for B in (300, 1, 2048, 77, 1500, 4):
        ops = np.array([tfhe.OPCODES[names[i]] for i in rng.integers(0, len(names), B)], np.uint8)
        ins = []
        for _ in range(3):
            a = tfhe.pinned_empty((B, K.params.lwe_size + 1))
            a[:] = tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data
            ins.append(a)
        jobs.append((ops, ins, eng80.gates(ops, *ins)))
    tickets, outs = [], []
    for ops, ins, _ in jobs:                      # never waited for explicitly until the end: submits 3.. displace the oldest
        t, o = eng80.gates_submit(ops, *ins)
        tickets.append(t); outs.append(o)
        assert t in (0, 1)
    for t in tickets:
        eng80.gates_wait(t)
    eng80.gates_wait(tickets[0])
    for (ops, ins, want), got in zip(jobs, outs):
        assert np.array_equal(got, want)
    with pytest.raises(tfhe.EngineError):
        eng80.gates_wait(7)
    # and interleaved with blocking calls on the same context
    t, o = eng80.gates_submit(jobs[2][0], *jobs[2][1])
    again = eng80.gates(jobs[0][0], *jobs[0][1])
    eng80.gates_wait(t)
    assert np.array_equal(o, jobs[2][2]) and np.array_equal(again, jobs[0][2])


@pytest.mark.parametrize("B", [5, 1027, 2049])
def test_lockstep_groups_ragged(tfhe, orc, keys80, eng80, B):
    """blind_rotate_kernel_v3<...,rw4> (four rotations per workgroup, one barrier every four CMUX steps; the default from
    2048 rotations up) on batch sizes that leave 3, 1 and 3 padding waves in the last workgroup: bit-equal to the oracle."""
    K = keys80
    rng = np.random.default_rng(B)
    ins = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data for _ in range(2)]
    ops = np.zeros(B, np.uint8)
    for name, value in (("br_small", -1), ("br_tiny", -1), ("v3_rw", 4), ("pipeline_min", -1)):
        eng80.set_option(name, value)
    try:
        got = eng80.gates(ops, *ins)
        assert eng80.last_kernel_name() == "blind_rotate_kernel_v3<2,8,tw2reg,rw4>"
    finally:
        for name, value in (("br_small", 1024), ("br_tiny", -2), ("v3_rw", 0), ("pipeline_min", 4096)):
            eng80.set_option(name, value)
    idx = np.unique(np.concatenate([np.arange(min(B, 8)), np.arange(max(0, B - 8), B)]))
    assert np.array_equal(got[idx], K.oracle.gates(ops[idx], ins[0][idx], ins[1][idx], nthreads=16))


def test_lwe_size_1023_and_what_is_still_refused(tfhe, orc):
    """lwe_size = 1023 (n + 1 = 1024 output words, the limit until round 4; larger sizes: tests/test_any_params.py).  Still refused,
    loudly: a polynomial degree that is no power of two (decode_message needs 2N to be one, numeric-functions.jl:28-33), N beyond
    8192 (UNSUPPORTED: one transform no longer fits a CU's LDS), multi-key with tlwe_mask_size != 1 (mk_internals.jl:89-91)."""
    from conftest import KeySet
    p = tfhe.SchemeParameters(1023, 1 / 2**17, 1024, 1, 2, 10, 9e-9, 8, 2, 1 / 2**17, 1)
    K = KeySet(tfhe, orc, p, seed=1023)
    eng = K.ck.engine(0)
    x = tfhe.encrypt(K.rng, K.sk, [True, False, True, False]).data
    y = tfhe.encrypt(K.rng, K.sk, [True, True, False, False]).data
    ops = np.zeros(4, np.uint8)
    got = eng.gates(ops, x, y)
    assert np.array_equal(got, K.oracle.gates(ops, x, y, nthreads=4))
    assert list(tfhe.decrypt(K.sk, got)) == [False, True, True, True]
    K.ck.close()
    for bad in ((10, 1000, 1, 2, 10, 8, 2, 1), (10, 16384, 1, 2, 10, 8, 2, 1), (10, 1024, 2, 2, 10, 8, 2, 2), (10, 1024, 1, 4, 9, 8, 2, 1), (10, 1024, 1, 2, 10, 8, 4, 1)):
        n, N, k, l, beta, t, gamma, parties = bad
        with pytest.raises(tfhe.EngineError):
            tfhe.Engine(tfhe.SchemeParameters(n, 0.0, N, k, l, beta, 0.0, t, gamma, 0.0, parties), 0)


def test_config3_full_size_mixed_stream(tfhe, orc, keys80, eng80):
    """BASELINE config 3 at full size on one GPU: 65 536 i.i.d. {NAND, AND, OR, XOR, MUX} gates (seed 789).
    Every output decrypts to the gate's truth value; rotation count = gates + MUXes; a contiguous shard computed alone equals
    the same rows of the full batch (what multi-GPU sharding relies on); 384 sampled rows of the full batch — 128 of them MUX,
    64 of every other opcode — equal the oracle word for word (gates.jl:15-177)."""
    from tfhe_jl_amd.sharding import shard_bounds
    K = keys80
    rng = np.random.default_rng(789)
    B = 65536
    names = ["NAND", "AND", "OR", "XOR", "MUX"]
    sel = rng.integers(0, 5, B)
    ops = np.array([tfhe.OPCODES[n] for n in names], np.uint8)[sel]
    bits = [rng.integers(0, 2, B).astype(bool) for _ in range(3)]
    ins = [tfhe.encrypt(K.rng, K.sk, b).data for b in bits]
    got = eng80.gates(ops, *ins)
    x, y, z = bits
    want = np.select([sel == 0, sel == 1, sel == 2, sel == 3, sel == 4], [~(x & y), x & y, x | y, x ^ y, np.where(x, y, z)])
    assert np.array_equal(tfhe.decrypt(K.sk, got), want)
    assert eng80.last_rotation_count() == B + int((sel == 4).sum())
    s, e = shard_bounds(ops, 8)[3]
    assert np.array_equal(eng80.gates(ops[s:e], *[a[s:e] for a in ins]), got[s:e])
    idx = np.concatenate([rng.choice(np.flatnonzero(sel == v), 128 if v == 4 else 64, replace=False) for v in range(5)])
    assert np.array_equal(got[idx], K.oracle.gates(ops[idx], *[a[idx] for a in ins], nthreads=orc.max_threads()))


def test_gpu_rounding_margin(tfhe, keys80, eng80, keys128, eng128):
    """The engine's own Float64 transform must stay far from a flipped rounding (assert the
    margin of the GPU FFT itself, not only the oracle's): max |pre-round value - nearest integer| < 0.25."""
    for K, eng in ((keys80, eng80), (keys128, eng128)):
        B = 256
        x = tfhe.encrypt(K.rng, K.sk, K.rng.integers(0, 2, B).astype(bool)).data
        y = tfhe.encrypt(K.rng, K.sk, K.rng.integers(0, 2, B).astype(bool)).data
        ops = np.zeros(B, np.uint8)
        ref = eng.gates(ops, x, y)
        eng.set_option("measure_margin", 1)
        got = eng.gates(ops, x, y)
        margin = eng.last_rounding_margin()
        eng.set_option("measure_margin", 0)
        assert np.array_equal(got, ref)
        assert 0.0 < margin < 0.25, margin


def test_nonstandard_keyswitch_shape_uses_fallback(tfhe, orc):
    """A keyswitch decomposition other than the shipped (t = 8, base 4) one — here t = 5, base 8 — takes the generic
    gather kernel (keyswitch_kernel) and must still match the oracle (keyswitch.jl:45-80 for any base / length)."""
    from conftest import KeySet
    p = tfhe.SchemeParameters(40, 1 / 2**15, 1024, 1, 2, 10, 9e-9, 5, 3, 1 / 2**15, 1)
    K = KeySet(tfhe, orc, p, seed=53)
    assert K.ck.keyswitch_key.shape == (1024, 5, 7, 41)
    eng = K.ck.engine(0)
    x = tfhe.encrypt(K.rng, K.sk, [True, False, True, False, True]).data
    y = tfhe.encrypt(K.rng, K.sk, [True, True, False, False, True]).data
    z = tfhe.encrypt(K.rng, K.sk, [False, True, True, False, False]).data
    for name in ("NAND", "MUX"):
        ops = np.full(5, tfhe.OPCODES[name], np.uint8)
        assert np.array_equal(eng.gates(ops, x, y, z), K.oracle.gates(ops, x, y, z, nthreads=5))
    K.ck.close()


def test_two_stream_host_batches_equal_one_stream(tfhe, keys80, eng80):
    """tfhe_gates_batch cuts large host-buffer batches in two rotation-balanced halves that run on two streams (the second
    half's upload and the first half's download overlap the other half's kernels).  Same words as the one-stream path, for
    a mixed stream with MUX (two rotations), NOT / CONSTANT / COPY (none), from pageable and from page-locked buffers, into
    a caller-supplied result array; the second stream's context follows a key reload."""
    K = keys80
    rng = np.random.default_rng(2024)
    B = 3000
    names = ["NAND", "AND", "OR", "XOR", "MUX", "NOT", "CONST1", "COPY", "XNOR"]
    ops = np.array([tfhe.OPCODES[n] for n in names], np.uint8)[rng.integers(0, len(names), B)]
    bits = [rng.integers(0, 2, B).astype(bool) for _ in range(3)]
    ins = [tfhe.encrypt(K.rng, K.sk, b).data for b in bits]
    eng80.set_option("pipeline_min", -1)
    want = eng80.gates(ops, *ins)
    rot = eng80.last_rotation_count()
    try:
        eng80.set_option("pipeline_min", 2)
        got = eng80.gates(ops, *ins)
        assert np.array_equal(got, want) and eng80.last_rotation_count() == rot
        pin = [tfhe.pinned_empty(a.shape) for a in ins]
        for p, a in zip(pin, ins):
            p[:] = a
        pout = tfhe.pinned_empty(want.shape)
        assert eng80.gates(ops, *pin, out=pout) is pout and np.array_equal(pout, want)
        # an operand array nobody reads may be absent; one that is read may not
        nand = np.zeros(64, np.uint8)
        assert np.array_equal(eng80.gates(nand, ins[0][:64], ins[1][:64]), eng80.gates(nand, ins[0][:64], ins[1][:64], ins[2][:64]))
        with pytest.raises(tfhe.EngineError):
            eng80.gates(ops[:64], ins[0][:64], ins[1][:64])          # the stream has MUXes: in2 is required
        # reload the keys: the second stream must use the new buffers
        eng80.load_bootstrap_key(K.ck.bootstrap_key)
        eng80.load_keyswitch_key(K.ck.keyswitch_key)
        assert np.array_equal(eng80.gates(ops, *ins), want)
    finally:
        eng80.set_option("pipeline_min", 4096)
