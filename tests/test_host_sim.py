"""The blind-rotate kernel's per-lane code (tfhe.jl_amd/csrc/br_core.hpp) executed on the CPU
(tests/host_sim/sim_br.cpp) and compared with the oracle.  No GPU involved: this checks the index
maths, LDS exchange layout, key layout, decomposition and rounding that the HIP kernel compiles from
the same header."""
import ctypes as C

import numpy as np
import pytest


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def test_bk_prepare_matches_reference_transform(host_sim, orc):
    rng = np.random.default_rng(7)
    polys = rng.integers(-2**31, 2**31, size=(3, 1024), dtype=np.int64).astype(np.int32)
    out = np.zeros((3, 8, 64, 2))
    host_sim.sim_bk_prepare(_p(polys), C.c_int64(3), _p(out))
    for q in range(3):
        ref = orc.forward_transform(polys[q]) / 512.0
        for lane in (0, 1, 9, 63):
            for k2 in range(8):
                f = host_sim.sim_freq_of(lane, k2)
                got = out[q, k2, lane, 0] + 1j * out[q, k2, lane, 1]
                assert abs(got - ref[f]) <= 1e-9 * max(1.0, abs(ref[f]))


@pytest.mark.parametrize("l,beta", [(2, 10), (3, 7)])
def test_blind_rotate_lane_code_matches_oracle(host_sim, orc, tfhe, l, beta):
    n, N = 12, 1024
    rng = np.random.default_rng(100 + l)
    from tfhe_jl_amd.keys import TLweKey, make_bootstrap_key
    from tfhe_jl_amd.lwe import LweKey
    lwe_key = LweKey(rng, n)
    tlwe_key = TLweKey(rng, N, 1)
    bk = make_bootstrap_key(rng, 1e-9, lwe_key, tlwe_key, l, beta)
    o = orc.Oracle(n, N, 1, l, beta, 8, 2)
    o.load_bootstrap_key(bk)
    spec = np.zeros((bk.size // N, 8, 64, 2))
    host_sim.sim_bk_prepare(_p(bk), C.c_int64(bk.size // N), _p(spec))
    mu = 2**29
    for trial in range(3):
        x = rng.integers(-2**31, 2**31, size=(1, n + 1), dtype=np.int64).astype(np.int32)
        if trial == 1:
            x[0, 3] = 5          # decode -> 0: the skipped step (bootstrap.jl:34)
            x[0, 4] = -2**31     # decode -> -N
        want = o.bootstrap(mu, x, with_keyswitch=False)[0]
        bara = np.array([orc.decode_message(int(v), 2 * N) for v in x[0]], np.int32)
        ext = np.zeros(N + 1, np.int32)
        margin = host_sim.sim_blind_rotate(_p(bara), C.c_int32(n), C.c_int32(l), C.c_int32(beta), C.c_int32(mu),
                                           _p(spec), _p(ext))
        assert np.array_equal(ext, want)
        assert margin < 0.25
        # the shipped kernel's lane code (blind_rotate_kernel_v3: folded twiddles, bit-field digits, no zero-skip)
        ext3 = np.zeros(N + 1, np.int32)
        margin3 = host_sim.sim_blind_rotate_v3(_p(bara), C.c_int32(n), C.c_int32(l), C.c_int32(beta), C.c_int32(mu),
                                               _p(spec), _p(ext3))
        assert np.array_equal(ext3, want)
        assert margin3 < 0.25
