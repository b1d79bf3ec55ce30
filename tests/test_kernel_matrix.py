"""Every blind-rotate kernel instantiation the dispatcher can select (launch_blind_rotate / tfhe_mk_gate_nand_batch in
csrc/engine_dispatch.hip, engine_multikey.hip), driven through the C ABI at an option setting or batch size that selects it, against the oracle
word for word; its DIAG instantiation must give the same words with a rounding margin far from a flipped rounding
(the reference rounds at polynomials.jl:115-116; its multi-key code sums products in Int32 precisely because it does
not trust spectrum-domain sums, mk_internals.jl:359-366 — here the margin of doing so is measured on the GPU).
BASELINE configs 4a / 4b / 5 at their stated batch sizes are in this file too."""
import numpy as np
import pytest

from conftest import DEVICE_PAIRS

pytestmark = pytest.mark.gpu

MU = 2**29
# decomposition (l, beta) per l: l * beta <= 32; beta = 10 only where the Float64 transform keeps its margin (l <= 2, N = 1024)
BETA_1024 = {1: 10, 2: 10, 3: 7, 4: 7}
BETA_OTHER = {1: 7, 2: 7, 3: 7, 4: 7}


def _setup(tfhe, orc, N, k, l, beta, n=10, seed=0):
    from conftest import KeySet
    p = tfhe.SchemeParameters(n, 1 / 2**15, N, k, l, beta, 9e-9, 8, 2, 1 / 2**15, 1)
    return KeySet(tfhe, orc, p, seed=1000 * N + 100 * k + 10 * l + seed)


def _words(rng, rows, width):
    x = rng.integers(-2**31, 2**31, size=(rows, width), dtype=np.int64).astype(np.int32)
    x[0, :] = 0
    x[1, :min(7, width)] = [2**31 - 1, -2**31, 2**20, 2**20 - 1, -2**20, -2**20 - 1, 1][:min(7, width)]
    return x


def _check_clock(eng, what):
    """The in-kernel clock of the DIAG run that just finished (tfhe_last_kernel_clock_mhz).  A launch that lasted a millisecond or
    more holds the sustained clock: 300 .. 2600 MHz on an MI355X.  The launches of this file's small cases last tens of
    microseconds: their reading may sit on the DVFS ramp of a device that just woke up (the 300 MHz bound failed once there —
    round-4 advice), and one whose workgroups ran for less than 10 us has no reading at all (TFHE_ERR_STATE): for those only
    that the figure, if there is one, is a plausible frequency."""
    import tfhe_jl_amd as tfhe
    long_launch = eng.last_timing_ms(0) >= 1.0
    try:
        clock = eng.last_kernel_clock_mhz()
    except tfhe.EngineError as e:
        assert e.code == 5 and not long_launch, (what, str(e))
        return None
    lo, hi = (300.0, 2600.0) if long_launch else (30.0, 3000.0)
    assert lo < clock < hi, (what, clock, long_launch)
    return clock


def _check(eng, K, x, expect_kernel, what):
    want = K.oracle.bootstrap(MU, x, with_keyswitch=False, nthreads=8)
    got = eng.bootstrap(MU, x, with_keyswitch=False)
    assert eng.last_kernel_name() == expect_kernel, (what, eng.last_kernel_name())
    assert np.array_equal(got, want), what
    eng.set_option("measure_margin", 1)
    try:
        again = eng.bootstrap(MU, x, with_keyswitch=False)
        assert eng.last_kernel_name() == expect_kernel
        margin = eng.last_rounding_margin()
        _check_clock(eng, what)
    finally:
        eng.set_option("measure_margin", 0)
    assert np.array_equal(again, want), what + " (DIAG instantiation)"
    assert 0.0 <= margin < 0.25, (what, margin)
    return margin


@pytest.mark.parametrize("l", [2, 3])
def test_single_key_kernels_n1024_tuned(tfhe, orc, l):
    """The tuned kernels of the shipped decomposition lengths (l = 2: tfhe_parameters_80, l = 3: tfhe_parameters_128), k = 1,
    N = 1024: blind_rotate_kernel_h2<l> (up to one rotation per CU), w2<l> (<= 1024 rotations), v3<l,8,tw2reg> with one and
    with four rotations per workgroup, and the switches between them by batch size."""
    K = _setup(tfhe, orc, 1024, 1, l, BETA_1024[l])
    eng = K.ck.engine(0)
    x = _words(np.random.default_rng(l), 6, K.params.lwe_size + 1)
    _check(eng, K, x, f"blind_rotate_kernel_h2<{l}>", f"h2<{l}>")    # the default for a batch this small
    eng.set_option("br_tiny", -1)
    _check(eng, K, x, f"blind_rotate_kernel_w2<{l}>", f"w2<{l}>")
    eng.set_option("br_small", -1)
    _check(eng, K, x, f"blind_rotate_kernel_v3<{l},8,tw2reg>", "v3 default")
    # four rotations per workgroup in lockstep (the default from 1536 rotations up): 6 rotations = one full group + one
    # with two padding waves, which recompute the last rotation and store nothing
    eng.set_option("v3_rw", 4)
    _check(eng, K, x, f"blind_rotate_kernel_v3<{l},8,tw2reg,rw4>", "v3 lockstep groups")
    eng.set_option("v3_rw", 0)
    eng.set_option("br_small", 1024)
    _check(eng, K, x, f"blind_rotate_kernel_w2<{l}>", f"w2<{l}>")
    # the switches between the kernels are by batch size, without any option: up to one rotation per CU (256 on an MI355X)
    # every transform is split over two waves, up to 1024 rotations a rotation takes two waves, beyond that one
    eng.set_option("br_tiny", -2)
    big = np.repeat(x[2:3], 1300, axis=0)
    big[:, 0] += np.arange(1300, dtype=np.int32) << 21       # distinct first exponents
    idx = [0, 1, 7, 8, 255, 256, 511, 512, 698, 699, 1023, 1024, 1299]
    want = K.oracle.bootstrap(MU, big[idx], with_keyswitch=False, nthreads=8)
    for rows, kernel in ((256, f"blind_rotate_kernel_h2<{l}>"), (257, f"blind_rotate_kernel_w2<{l},rw2>"),
                         (700, f"blind_rotate_kernel_w2<{l}>"), (1024, f"blind_rotate_kernel_w2<{l},rw2>"),
                         (1025, f"blind_rotate_kernel_v3<{l},8,tw2reg>"), (1300, f"blind_rotate_kernel_v3<{l},8,tw2reg>")):
        got = eng.bootstrap(MU, big[:rows], with_keyswitch=False)
        assert eng.last_kernel_name() == kernel, (rows, eng.last_kernel_name())
        sel = [j for j, r in enumerate(idx) if r < rows]
        assert np.array_equal(got[[idx[j] for j in sel]], want[sel]), rows
    K.ck.close()


@pytest.mark.parametrize("N,k,l", [(1024, 2, 1), (1024, 2, 4), (2048, 1, 1), (2048, 1, 2), (2048, 1, 4)])
def test_unshipped_decomposition_lengths_take_the_general_kernel(tfhe, orc, N, k, l):
    """l = 1 and l = 4 (and l = 2 at N = 2048) belong to no shipped parameter set: their tuned instantiations of round 3 are
    gone (build weight) and, for k = 2 or N = 2048, blind_rotate_kernel_general runs them — same words as the oracle, margin
    asserted."""
    K = _setup(tfhe, orc, N, k, l, BETA_OTHER[l], n=8)
    eng = K.ck.engine(0)
    x = _words(np.random.default_rng(60 + l), 5, K.params.lwe_size + 1)
    _check(eng, K, x, f"blind_rotate_kernel_general(N={N},k={k},l={l})", "general")
    K.ck.close()


@pytest.mark.parametrize("l,beta", [(1, 10), (4, 8), (5, 6), (8, 4)])
def test_any_decomposition_length_on_the_tuned_kernels(tfhe, orc, l, beta):
    """k = 1, N = 1024 with a decomposition length no shipped set uses: the one- and two-waves-per-rotation kernels instantiated
    with L = 0 read l at run time (their transform loops are rolled; nothing else depends on it), so such a set runs at the speed
    of the tuned ones instead of on the general kernel.  Words against the oracle, DIAG margin, names; and the switch by batch
    size (no 4 l-wave kernel for these: the two-wave kernel takes the smallest batches as well)."""
    K = _setup(tfhe, orc, 1024, 1, l, beta, n=8)
    eng = K.ck.engine(0)
    x = _words(np.random.default_rng(80 + l), 6, K.params.lwe_size + 1)
    _check(eng, K, x, f"blind_rotate_kernel_w2<0>(l={l})", "w2, run-time l")
    eng.set_option("br_small", -1)
    _check(eng, K, x, f"blind_rotate_kernel_v3<0,8,tw2reg>(l={l})", "v3, run-time l")
    eng.set_option("v3_rw", 4)
    _check(eng, K, x, f"blind_rotate_kernel_v3<0,8,tw2reg,rw4>(l={l})", "v3 lockstep groups, run-time l")
    eng.set_option("v3_rw", 0)
    eng.set_option("br_small", 1024)
    eng.set_option("br_general", 1)
    _check(eng, K, x, f"blind_rotate_kernel_general(N=1024,k=1,l={l})", "the general kernel on the same set")
    eng.set_option("br_general", 0)
    if l == 4:
        # a batch of whole rounds + a small tail: the split launch, with the two-wave kernel (not the 4 l-wave one) for the tail
        big = np.repeat(x[2:3], 2100, axis=0)
        big[:, 0] += np.arange(2100, dtype=np.int32) << 20
        idx = [0, 1, 2047, 2048, 2099]
        got = eng.bootstrap(MU, big, with_keyswitch=False)
        assert eng.last_kernel_name() == "blind_rotate_kernel_v3<0,8,tw2reg,rw4>(l=4) + blind_rotate_kernel_w2<0>(l=4)", eng.last_kernel_name()
        assert np.array_equal(got[idx], K.oracle.bootstrap(MU, big[idx], with_keyswitch=False, nthreads=8))
    K.ck.close()


@pytest.mark.parametrize("l", [2, 3])
def test_run_time_l_instantiations_equal_the_templated_ones(tfhe, orc, l):
    """Option br_rt_l: the L = 0 instantiations on the shipped decomposition lengths give the words of the <2> / <3> ones."""
    K = _setup(tfhe, orc, 1024, 1, l, BETA_1024[l])
    eng = K.ck.engine(0)
    x = _words(np.random.default_rng(90 + l), 6, K.params.lwe_size + 1)
    eng.set_option("br_tiny", -1)
    want = eng.bootstrap(MU, x, with_keyswitch=False)
    assert eng.last_kernel_name() == f"blind_rotate_kernel_w2<{l}>"
    eng.set_option("br_rt_l", 1)
    _check(eng, K, x, f"blind_rotate_kernel_w2<0>(l={l})", "w2<0>")
    assert np.array_equal(eng.bootstrap(MU, x, with_keyswitch=False), want)
    eng.set_option("br_small", -1)
    _check(eng, K, x, f"blind_rotate_kernel_v3<0,8,tw2reg>(l={l})", "v3<0>")
    assert np.array_equal(eng.bootstrap(MU, x, with_keyswitch=False), want)
    K.ck.close()


@pytest.mark.parametrize("l", [2, 3])
def test_mask_size_2_kernel(tfhe, orc, l):
    """tlwe_mask_size = 2 (api.jl:30,55), l = 2 and 3 (the shipped sets with that keyword): blind_rotate_kernel_k2w3<l> (three waves
    per rotation, wave c owns polynomial c: the default up to two rotations per CU, round 5) and blind_rotate_kernel_k2<l> (one wave
    per rotation; alone or in lockstep groups of up to seven per workgroup), each with its DIAG instantiation."""
    K = _setup(tfhe, orc, 1024, 2, l, BETA_OTHER[l], n=8)
    eng = K.ck.engine(0)
    x = _words(np.random.default_rng(20 + l), 5, K.params.lwe_size + 1)
    _check(eng, K, x, f"blind_rotate_kernel_k2w3<{l}>", f"k2w3<{l}>")     # the default for a batch this small
    eng.set_option("k2_w3", 0)
    eng.set_option("k2_rw", 1)                # single-rotation workgroups (and their DIAG instantiation)
    _check(eng, K, x, f"blind_rotate_kernel_k2<{l}>", f"k2<{l}>")
    # up to seven rotations per workgroup in lockstep, dealt out in equally full rounds of one workgroup per
    # CU — 5 rotations = five workgroups of one rotation and six idle waves each, then 9
    eng.set_option("k2_rw", 0)
    got = eng.bootstrap(MU, x, with_keyswitch=False)
    assert eng.last_kernel_name() == f"blind_rotate_kernel_k2<{l},rw7>"
    assert np.array_equal(got, K.oracle.bootstrap(MU, x, with_keyswitch=False, nthreads=8))
    x9 = _words(np.random.default_rng(30 + l), 9, K.params.lwe_size + 1)
    want9 = K.oracle.bootstrap(MU, x9, with_keyswitch=False, nthreads=8)
    assert np.array_equal(eng.bootstrap(MU, x9, with_keyswitch=False), want9)
    eng.set_option("k2_w3", 1)                # the three-wave kernel whatever the batch size
    assert np.array_equal(eng.bootstrap(MU, x9, with_keyswitch=False), want9)
    assert eng.last_kernel_name() == f"blind_rotate_kernel_k2w3<{l}>"
    eng.set_option("k2_w3", -1)
    K.ck.close()


def test_mask_size_2_balanced_rounds(tfhe, orc):
    """The k = 2 dispatcher rules on a 256-CU device, 1792 + 300 rotations.  Without the three-wave kernel (k2_w3 = 0): two equally
    full rounds of the lockstep groups (workgroups of 5 and 4 rotations, the other waves of a group idling at the barriers).  By
    default (round 5): one full round of seven per CU, the 300 left over on blind_rotate_kernel_k2w3.  Rows from the first, the
    last and the boundary workgroups equal the oracle either way."""
    K = _setup(tfhe, orc, 1024, 2, 2, BETA_OTHER[2], n=8)
    eng = K.ck.engine(0)
    R = 1792 + 300
    x = np.repeat(_words(np.random.default_rng(77), 4, K.params.lwe_size + 1), (R + 3) // 4, axis=0)[:R].copy()
    x[:, 0] += (np.arange(R, dtype=np.int64) << 20).astype(np.int32)          # distinct first exponents
    idx = [0, 1, 4, 5, 6, 219, 220, 221, 224, 225, 1791, 1792, 1793, 2000, R - 2, R - 1]
    want = K.oracle.bootstrap(MU, x[idx], with_keyswitch=False, nthreads=8)
    got = eng.bootstrap(MU, x, with_keyswitch=False)
    assert eng.last_kernel_name() == "blind_rotate_kernel_k2<2,rw7> + blind_rotate_kernel_k2w3<2>", eng.last_kernel_name()
    assert np.array_equal(got[idx], want)
    eng.set_option("k2_w3", 0)
    got0 = eng.bootstrap(MU, x, with_keyswitch=False)
    assert eng.last_kernel_name() == "blind_rotate_kernel_k2<2,rw7>"
    assert np.array_equal(got0, got)
    K.ck.close()


def test_n2048_kernel(tfhe, orc):
    """blind_rotate_kernel_n2048x<3> (synthetic N = 2048, BASELINE config 4b's shape: l = 3, beta = 7): the rotated words of a
    polynomial are computed by one wave and handed to the other (round 4); one and two rotations per workgroup (5 rotations =
    a padded group)."""
    l = 3
    K = _setup(tfhe, orc, 2048, 1, l, BETA_OTHER[l], n=8)
    eng = K.ck.engine(0)
    x = _words(np.random.default_rng(40 + l), 5, K.params.lwe_size + 1)
    _check(eng, K, x, f"blind_rotate_kernel_n2048x<{l},rw1>", f"n2048x<{l}>")
    eng.set_option("n2048_rw", 2)            # (the DIAG instantiation exists for single-rotation workgroups only)
    got = eng.bootstrap(MU, x, with_keyswitch=False)
    assert eng.last_kernel_name() == f"blind_rotate_kernel_n2048x<{l},rw2>"
    assert np.array_equal(got, K.oracle.bootstrap(MU, x, with_keyswitch=False, nthreads=8))
    K.ck.close()


def test_pair_handoffs_across_the_meeting_barrier(tfhe, orc):
    """Round 6: with two rotations per workgroup the hand-offs of blind_rotate_kernel_n2048x and mk_blind_rotate_kernel_w2 synchronise the two
    waves they concern through polled LDS words (pair_signal, kernels_common.hpp) and the rotations meet at a workgroup barrier every 32
    steps.  The small cases of this file have 8 - 12 steps: here 70 (N = 2048) and 2 x 40 (two parties) — across the meeting barrier twice —,
    an odd number of rotations (the last workgroup holds a padding rotation, which must keep every appointment of its pair and of its
    workgroup) and two rotations per workgroup forced on a batch the dispatcher would run one by one."""
    K = _setup(tfhe, orc, 2048, 1, 3, BETA_OTHER[3], n=70, seed=6)
    eng = K.ck.engine(0)
    x = _words(np.random.default_rng(406), 7, K.params.lwe_size + 1)
    eng.set_option("n2048_rw", 2)
    got = eng.bootstrap(MU, x, with_keyswitch=False)
    assert eng.last_kernel_name() == "blind_rotate_kernel_n2048x<3,rw2>"
    assert np.array_equal(got, K.oracle.bootstrap(MU, x, with_keyswitch=False, nthreads=8))
    K.ck.close()
    p, rng, sks, ck, o = _mk(tfhe, orc, 2, 4, 7, 40, 2, 606)
    eng = ck.engine(0)
    x, y = _words(rng, 7, 2 * 40 + 1), _words(rng, 7, 2 * 40 + 1)[::-1].copy()
    x[2:4] = tfhe.mk_encrypt(rng, sks, [True, False])
    y[2:4] = tfhe.mk_encrypt(rng, sks, [True, True])
    eng.set_option("mk_rw", 2)
    got = eng.mk_gate_nand(x, y)
    assert eng.last_kernel_name() == "mk_blind_rotate_kernel_w2<4>"
    assert np.array_equal(got, o.mk_gate_nand(x, y, nthreads=8))
    ck.close()


GENERAL = [  # what, N, k, l, beta        parameter sets the reference accepts (api.jl:4-21,30,55) and no specialised kernel covers
    ("tfhe_parameters_80(tlwe_mask_size=3)", 1024, 3, 2, 10),
    ("tlwe_mask_size=4", 1024, 4, 2, 8),
    ("single key, l = 5 / beta = 6 (br_general: by default the run-time-l kernels take it)", 1024, 1, 5, 6),
    ("k = 2, l = 6 / beta = 5", 1024, 2, 6, 5),
    ("N = 2048, k = 2", 2048, 2, 3, 7),
    ("N = 2048, k = 1, l = 5", 2048, 1, 5, 5),
]


@pytest.mark.parametrize("what,N,k,l,beta", GENERAL, ids=[g[0] for g in GENERAL])
def test_general_single_key_kernel(tfhe, orc, what, N, k, l, beta):
    """blind_rotate_kernel_general: never TFHE_ERR_UNSUPPORTED for a set the reference would run (round-3 verdict, missing #3).
    Arbitrary words through the blind rotation + its DIAG instantiation (margin < 0.25), then NAND and MUX truth tables
    through the whole gate (keyswitch from kN = k N words) against the oracle word for word, and decrypted."""
    import itertools
    K = _setup(tfhe, orc, N, k, l, beta, n=6)
    eng = K.ck.engine(0)
    if (N, k) == (1024, 1):
        eng.set_option("br_general", 1)
    x = _words(np.random.default_rng(7 * N + k + l), 5, K.params.lwe_size + 1)
    _check(eng, K, x, f"blind_rotate_kernel_general(N={N},k={k},l={l})", what)
    combos = list(itertools.product((False, True), repeat=3))
    ins = [tfhe.encrypt(K.rng, K.sk, [c[i] for c in combos]).data for i in range(3)]
    for name, ref in (("NAND", lambda a, b, c: not (a and b)), ("MUX", lambda a, b, c: b if a else c)):
        ops = np.full(8, tfhe.OPCODES[name], np.uint8)
        got = eng.gates(ops, *ins)
        assert np.array_equal(got, K.oracle.gates(ops, *ins, nthreads=8)), (what, name)
        assert list(tfhe.decrypt(K.sk, got)) == [bool(ref(*c)) for c in combos], (what, name)
    K.ck.close()


@pytest.mark.parametrize("N,k,l", [(1024, 1, 2), (1024, 2, 2), (2048, 1, 3)])
def test_general_kernel_equals_the_specialised_ones(tfhe, orc, N, k, l):
    """Option br_general: the fallback kernel on sets that have a tuned kernel gives the same words (and the oracle's)."""
    K = _setup(tfhe, orc, N, k, l, BETA_1024[l] if N == 1024 else BETA_OTHER[l], n=6)
    eng = K.ck.engine(0)
    x = _words(np.random.default_rng(3), 4, K.params.lwe_size + 1)
    tuned = eng.bootstrap(MU, x, with_keyswitch=False)
    assert "general" not in eng.last_kernel_name()
    eng.set_option("br_general", 1)
    _check(eng, K, x, f"blind_rotate_kernel_general(N={N},k={k},l={l})", "br_general")
    assert np.array_equal(eng.bootstrap(MU, x, with_keyswitch=False), tuned)
    K.ck.close()


def _mk(tfhe, orc, parties, l, beta, n, max_parties, seed):
    p = tfhe.SchemeParameters(n, 0.012467, 1024, 1, l, beta, 3.29e-10, 8, 2, 2.44e-5, max_parties)
    rng = np.random.default_rng(seed)
    sks = [tfhe.SecretKey(rng, p) for _ in range(parties)]
    shared = tfhe.SharedKey(rng, p)
    ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, sk, shared) for sk in sks])
    o = orc.Oracle(n, 1024, 1, l, beta, 8, 2, parties=parties)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    return p, rng, sks, ck, o


def _mk_check(eng, o, x, y, expect_kernel):
    want = o.mk_gate_nand(x, y, nthreads=8)
    got = eng.mk_gate_nand(x, y)
    assert eng.last_kernel_name() == expect_kernel, eng.last_kernel_name()
    assert np.array_equal(got, want)
    eng.set_option("measure_margin", 1)
    try:
        again = eng.mk_gate_nand(x, y)
        margin = eng.last_rounding_margin()
        _check_clock(eng, expect_kernel)
    finally:
        eng.set_option("measure_margin", 0)
    assert np.array_equal(again, want)
    assert 0.0 <= margin < 0.25, margin
    return margin


@pytest.mark.parametrize("l", [2, 3, 4])
def test_mk_two_party_kernels(tfhe, orc, l):
    """2 parties.  l = 4, beta = 7 is mktfhe_parameters_2party (mk_api.jl:4-10): mk_blind_rotate_kernel_w2<4> (two waves per
    rotation) with the any-party kernel as its cross-check (option mk_general).  Other decomposition lengths belong to no
    shipped multi-key set and take the any-party kernel."""
    p, rng, sks, ck, o = _mk(tfhe, orc, 2, l, 7, 12, 2, 60 + l)
    eng = ck.engine(0)
    x, y = _words(rng, 5, 2 * 12 + 1), _words(rng, 5, 2 * 12 + 1)[::-1].copy()
    x[2:4] = tfhe.mk_encrypt(rng, sks, [True, False])
    y[2:4] = tfhe.mk_encrypt(rng, sks, [True, True])
    if l == 4:
        _mk_check(eng, o, x, y, "mk_blind_rotate_kernel_w2<4>")
        eng.set_option("mk_general", 1)
        _mk_check(eng, o, x, y, "mk_blind_rotate_kernel_general(P=2,L=4)")
    else:
        _mk_check(eng, o, x, y, f"mk_blind_rotate_kernel_general(P=2,L={l})")
    ck.close()


@pytest.mark.parametrize("which,parties,l,beta,n", [("4party", 4, 5, 6, 8), ("8party", 8, 8, 4, 4), ("3-of-4", 3, 5, 6, 8), ("2party-general", 2, 4, 7, 12)])
def test_mk_general_kernel_margin(tfhe, orc, which, parties, l, beta, n):
    """mk_blind_rotate_kernel_g2<P,l> and mk_blind_rotate_kernel_general at the shipped 4- and 8-party decompositions (mk_api.jl:16-34: l = 5 / beta = 6,
    l = 8 / beta = 4): up to (P+1) l products are summed in the spectrum domain before ONE rounding — the margin
    of exactly that is asserted on the GPU."""
    p, rng, sks, ck, o = _mk(tfhe, orc, parties, l, beta, n, max(parties, 2) if which != "3-of-4" else 4, 80 + parties)
    eng = ck.engine(0)
    if parties == 2:
        eng.set_option("mk_general", 1)
    w = parties * n + 1
    x, y = _words(rng, 4, w), _words(rng, 4, w)[::-1].copy()
    x[2:4] = tfhe.mk_encrypt(rng, sks, [True, False])
    y[2:4] = tfhe.mk_encrypt(rng, sks, [True, True])
    if which in ("4party", "8party"):
        # default for the shipped shapes: two waves per rotation, compile-time (parties, l); DIAG instantiation included
        # (4 parties: the accumulator images in LDS, round 4; 8 parties: in global memory)
        _mk_check(eng, o, x, y, f"mk_blind_rotate_kernel_g2<{parties},{l}" + (",acc=lds>" if parties == 4 else ">"))
        eng.set_option("mkg_variant", 1)
    _mk_check(eng, o, x, y, f"mk_blind_rotate_kernel_general(P={parties},L={l}" + (",acc=global)" if parties > 4 else ")"))
    ck.close()


# ---- BASELINE configurations at their stated batch sizes ----------------------------------------------------------
def test_config4a_128bit_4096(tfhe, orc, keys128):
    """BASELINE config 4a: tfhe_parameters_128 (api.jl:55-69), 4096 NAND on one GPU -> blind_rotate_kernel_v3<3,8,tw2reg,rw4>.
    Every output decrypts to NAND; 256 sampled rows equal the oracle word for word; DIAG run identical, margin < 0.25."""
    K = keys128
    eng = K.ck.engine(0)
    rng = np.random.default_rng(4128)
    B = 4096
    bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.encrypt(K.rng, K.sk, bx).data, tfhe.encrypt(K.rng, K.sk, by).data
    ops = np.zeros(B, np.uint8)
    got = eng.gates(ops, x, y)
    assert eng.last_kernel_name() == "blind_rotate_kernel_v3<3,8,tw2reg,rw4>"
    assert np.array_equal(tfhe.decrypt(K.sk, got), ~(bx & by))
    idx = rng.choice(B, 256, replace=False)            # >= 256 sampled rows against the oracle
    assert np.array_equal(got[idx], K.oracle.gates(ops[idx], x[idx], y[idx], nthreads=orc.max_threads()))
    eng.set_option("measure_margin", 1)
    again = eng.gates(ops, x, y)
    margin = eng.last_rounding_margin()
    eng.set_option("measure_margin", 0)
    assert np.array_equal(again, got) and 0.0 < margin < 0.25, margin


def test_config4b_synthetic_n2048_4096(tfhe, orc):
    """BASELINE config 4b: synthetic N = 2048 (n = 630, l = 3, beta = 7), 4096 NAND -> blind_rotate_kernel_n2048x<3,rw2>."""
    from conftest import KeySet
    from test_oracle import synthetic_2048
    K = KeySet(tfhe, orc, synthetic_2048(tfhe), seed=2048)
    eng = K.ck.engine(0)
    rng = np.random.default_rng(4204)
    B = 4096
    bx, by = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.encrypt(K.rng, K.sk, bx).data, tfhe.encrypt(K.rng, K.sk, by).data
    ops = np.zeros(B, np.uint8)
    got = eng.gates(ops, x, y)
    assert eng.last_kernel_name() == "blind_rotate_kernel_n2048x<3,rw2>"
    assert np.array_equal(tfhe.decrypt(K.sk, got), ~(bx & by))
    idx = rng.choice(B, 256, replace=False)            # >= 256 sampled rows against the oracle
    assert np.array_equal(got[idx], K.oracle.gates(ops[idx], x[idx], y[idx], nthreads=orc.max_threads()))
    for _ in range(2):                               # the same full launch again: every word identical (a race between the two
        assert np.array_equal(eng.gates(ops, x, y), got)   # waves of a rotation would show as a run-to-run difference)
    eng.set_option("br_prio_pct", 0)                 # without the issue-priority schedule: same words
    assert np.array_equal(eng.gates(ops, x, y), got)
    eng.set_option("br_general", 1)                  # the any-parameter kernel on a sample: same words
    assert np.array_equal(eng.gates(ops[:32], x[:32], y[:32]), got[:32])
    assert eng.last_kernel_name() == "blind_rotate_kernel_general(N=2048,k=1,l=3)"
    eng.set_option("br_prio_pct", 90)
    eng.set_option("measure_margin", 1)
    again = eng.gates(ops[:256], x[:256], y[:256])
    margin = eng.last_rounding_margin()
    eng.set_option("measure_margin", 0)
    assert np.array_equal(again, got[:256]) and 0.0 < margin < 0.25, margin
    K.ck.close()


def test_config5_mk_two_party_1024(tfhe, orc):
    """BASELINE config 5: mktfhe_parameters_2party (mk_api.jl:4-10), 1024 NAND -> mk_blind_rotate_kernel_w2<4>.
    256 sampled rows equal the oracle word for word (decrypt-level MK checks are ~0.2 %/gate noisy by design of the
    scheme's parameters, SURVEY §4: at least 98.5 % must decrypt to NAND)."""
    p = tfhe.mktfhe_parameters_2party
    rng = np.random.default_rng(321)
    sks = [tfhe.SecretKey(rng, p) for _ in range(2)]
    shared = tfhe.SharedKey(rng, p)
    ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, sk, shared) for sk in sks])
    o = orc.Oracle(p.lwe_size, 1024, 1, p.bs_decomp_length, p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base, parties=2)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    eng = ck.engine(0)
    B = 1024
    m1, m2 = rng.integers(0, 2, B).astype(bool), rng.integers(0, 2, B).astype(bool)
    x, y = tfhe.mk_encrypt(rng, sks, m1), tfhe.mk_encrypt(rng, sks, m2)
    got = eng.mk_gate_nand(x, y)
    assert eng.last_kernel_name() == "mk_blind_rotate_kernel_w2<4>"
    assert (tfhe.mk_decrypt(sks, got) == ~(m1 & m2)).mean() >= 0.985
    idx = rng.choice(B, 256, replace=False)            # >= 256 sampled rows against the oracle
    assert np.array_equal(got[idx], o.mk_gate_nand(x[idx], y[idx], nthreads=orc.max_threads()))
    for _ in range(2):                               # the same full launch again: every word identical (no race between the waves)
        assert np.array_equal(eng.mk_gate_nand(x, y), got)
    eng.set_option("br_prio_pct", 0)                 # without the issue-priority schedule: same words
    assert np.array_equal(eng.mk_gate_nand(x, y), got)
    eng.set_option("br_prio_pct", 90)
    eng.set_option("measure_margin", 1)
    again = eng.mk_gate_nand(x[:128], y[:128])
    margin = eng.last_rounding_margin()
    eng.set_option("measure_margin", 0)
    assert np.array_equal(again, got[:128]) and 0.0 < margin < 0.25, margin
    eng.set_option("mk_general", 1)                 # the any-party kernel gives the same words
    assert np.array_equal(eng.mk_gate_nand(x[:96], y[:96]), got[:96]) and eng.last_kernel_name() == "mk_blind_rotate_kernel_general(P=2,L=4)"
    eng.set_option("mk_general", 0)
    # the fan-out context (two device contexts on this one GPU; on two different GPUs where the box has them) gives the same words
    from conftest import device_count
    for devs in ([0, 0], [0, 1])[:2 if device_count() >= 2 else 1]:
        e2 = ck.engine(devs)
        assert e2.device_count() == 2
        assert np.array_equal(e2.mk_gate_nand(x[:65], y[:65]), got[:65]), devs
    ck.close()


# ---- the multi-device context behind the ABI ----------------------------------------------------------
@pytest.mark.parametrize("devs", DEVICE_PAIRS)
def test_multi_device_context_equals_single(tfhe, orc, keys80, devs):
    """tfhe_ctx_create_multi with device_ids = {0, 0}: keys replicated, a mixed batch split into rotation-balanced
    shards run concurrently, results written into the caller's buffer — identical to the one-device context."""
    K = keys80
    e1 = K.ck.engine(0)
    e2 = K.ck.engine(devs)
    assert e2.device_count() == 2 and e1.device_count() == 1
    rng = np.random.default_rng(8)
    names = ["NAND", "AND", "OR", "XOR", "MUX", "NOT", "CONST1", "COPY"]
    for B in (1, 2, 3, 97, 1500):
        ops = np.array([tfhe.OPCODES[names[i]] for i in rng.integers(0, len(names), B)], np.uint8)
        ins = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data for _ in range(3)]
        want = e1.gates(ops, *ins)
        rot = e1.last_rotation_count()
        got = e2.gates(ops, *ins)
        assert np.array_equal(got, want), B
        assert e2.last_rotation_count() == rot
        assert e2.last_timing_ms(2) >= 0.0
    x = _words(rng, 9, 501)
    ext = e1.bootstrap(MU, x, with_keyswitch=False)
    assert np.array_equal(e2.bootstrap(MU, x, with_keyswitch=False), ext)
    assert np.array_equal(e2.bootstrap(MU, x), e1.bootstrap(MU, x))
    assert np.array_equal(e2.keyswitch(ext), e1.keyswitch(ext))
    assert np.array_equal(e2.gates(np.zeros(0, np.uint8), np.zeros((0, 501), np.int32)), np.zeros((0, 501), np.int32))
    # the streaming entry points: every device takes its shard as a submit of its own (round 4); the result is complete after the wait
    ops = np.zeros(300, np.uint8)
    ins = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, 300).astype(bool)).data for _ in range(2)]
    t, o = e2.gates_submit(ops, *ins)
    assert t in (0, 1)
    e2.gates_wait(t)
    assert np.array_equal(o, e1.gates(ops, *ins))
    e2.gates_wait(t)                                                            # waiting twice is harmless
    with pytest.raises(tfhe.EngineError):
        e2.gates(np.array([0, 99, 0], np.uint8), x[:3], x[:3])                  # bad opcode: nothing runs
    with pytest.raises(tfhe.EngineError):
        e2.gates_dev(np.zeros(1, np.uint8), 1, 1, 1, 1, 1)                      # device pointers need one device
    # diagnostics and options reach every device context
    e2.set_option("measure_margin", 1)
    ops = np.zeros(64, np.uint8)
    ins = [tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, 64).astype(bool)).data for _ in range(2)]
    got = e2.gates(ops, *ins)
    assert 0.0 < e2.last_rounding_margin() < 0.25
    e2.set_option("measure_margin", 0)
    assert np.array_equal(got, e1.gates(ops, *ins))
    with pytest.raises(tfhe.EngineError):
        tfhe.Engine(K.params, devices=[0, 99])                                   # one bad id fails the whole create


class _Hip:
    """The few HIP runtime calls the stream test needs, through ctypes on the runtime the engine already loaded
    (importing torch here would bring a second HIP runtime into the process)."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.lib = C.CDLL("libamdhip64.so")
        for f in ("hipStreamCreate", "hipStreamDestroy", "hipStreamSynchronize", "hipMalloc", "hipFree", "hipMemcpy", "hipDeviceSynchronize", "hipSetDevice"):
            getattr(self.lib, f).restype = C.c_int

    def ok(self, rc):
        assert rc == 0, f"HIP error {rc}"

    def stream(self):
        h = self.C.c_void_p()
        self.ok(self.lib.hipStreamCreate(self.C.byref(h)))
        return h

    def upload(self, a):
        p = self.C.c_void_p()
        self.ok(self.lib.hipMalloc(self.C.byref(p), self.C.c_size_t(a.nbytes)))
        self.ok(self.lib.hipMemcpy(p, a.ctypes.data_as(self.C.c_void_p), self.C.c_size_t(a.nbytes), 1))
        return p

    def alloc(self, nbytes):
        p = self.C.c_void_p()
        self.ok(self.lib.hipMalloc(self.C.byref(p), self.C.c_size_t(nbytes)))
        return p

    def download(self, p, shape):
        out = np.empty(shape, np.int32)
        self.ok(self.lib.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), p, self.C.c_size_t(out.nbytes), 2))
        return out


def test_alternating_streams_share_one_context(tfhe, orc, keys80):
    """Batch calls issued on two different HIP streams against ONE context (shared workspaces): each call waits, on the
    device, for the context-owned event the previous call recorded — no host block, no foreign stream handle kept.
    One of the streams is synchronised, DESTROYED and replaced between calls (legal for the caller; a stored handle
    would dangle).  All outputs equal the oracle."""
    K = keys80
    eng = K.ck.engine(0)
    hip = _Hip()
    hip.ok(hip.lib.hipSetDevice(0))
    rng = np.random.default_rng(77)
    B = 600                     # > 512 rotations: the one-wave-per-rotation kernel, ~2 ms per call
    ops = np.zeros(B, np.uint8)
    streams = [hip.stream(), hip.stream()]
    batches = []
    for it in range(6):
        hx = tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data
        hy = tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, B).astype(bool)).data
        dx, dy, dout = hip.upload(hx), hip.upload(hy), hip.alloc(B * 501 * 4)
        s = streams[it & 1]
        eng.gates_dev(ops, dx.value, dy.value, 0, dout.value, B, s.value)     # returns before the kernels finish
        batches.append((hx, hy, dx, dy, dout))
        if it == 3:             # the caller may destroy a stream it has synchronised; the context must not care
            hip.ok(hip.lib.hipStreamSynchronize(streams[1]))
            hip.ok(hip.lib.hipStreamDestroy(streams[1]))
            streams[1] = hip.stream()
    hip.ok(hip.lib.hipDeviceSynchronize())
    idx = [0, 1, B // 2, B - 1]
    for hx, hy, dx, dy, dout in batches:
        got = hip.download(dout, (B, 501))
        assert np.array_equal(got[idx], K.oracle.gates(ops[idx], hx[idx], hy[idx], nthreads=4))
        for p in (dx, dy, dout):
            hip.ok(hip.lib.hipFree(p))
    for s in streams:
        hip.ok(hip.lib.hipStreamDestroy(s))


def test_wires_gather_any_order(tfhe, keys80):
    K = keys80
    eng = K.ck.engine(0)
    rng = np.random.default_rng(3)
    m = rng.integers(-2**31, 2**31, size=(40, 501), dtype=np.int64).astype(np.int32)
    eng.wires_alloc(40)
    eng.wires_upload(0, m)
    idx = [39, 0, 7, 7, 21]
    assert np.array_equal(eng.wires_gather(idx), m[idx])
    assert eng.wires_gather([]).shape == (0, 501)
    with pytest.raises(tfhe.EngineError):
        eng.wires_gather([40])
    eng.wires_alloc(0)
