"""Is a parameter set inside what a Float64 transform computes exactly?  tfhe_ctx_create decides from the parameters alone
(csrc/engine_context.hip: exactness_class; tfhe_get_option "exact_domain") and the host-side mirror SchemeParameters.exactness()
states the same law — the reference only warns (src/polynomials.jl:135-144).  2 = exact for ANY Int32 key words, 1 = exact for every
real (uniform) key, 0 = outside (the Python and Julia constructors warn once)."""
import math
import os
import re
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shipped_sets_and_the_law(tfhe):
    # every parameter set the reference ships (api.jl:30-69, mk_api.jl:4-34) and BASELINE config 4b's synthetic N = 2048 set
    c80, b80, m80 = tfhe.tfhe_parameters_80().exactness()
    assert c80 == 1 and b80 == 52.0 and 0.05 < m80 < 0.12         # all-keys bound exactly 2^52: one bit above the rounding trick's 2^51
    for p in (tfhe.tfhe_parameters_128(), tfhe.mktfhe_parameters_2party, tfhe.mktfhe_parameters_4party, tfhe.mktfhe_parameters_8party,
              tfhe.SchemeParameters(630, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)):
        c, b, m = p.exactness()
        assert c == 2 and b < 51 and m < 0.03, (p, c, b, m)
    assert tfhe.tfhe_parameters_80(tlwe_mask_size=2).exactness()[0] == 1
    # the measured margins of rounds 2-5 (DIAG instantiations) lie below the prediction
    measured = {(1024, 1, 2, 10, 1): 0.078, (1024, 1, 3, 7, 1): 0.012, (2048, 1, 3, 7, 1): 0.020, (1024, 2, 2, 10, 1): 0.094, (512, 1, 2, 10, 1): 0.027,
                (4096, 1, 3, 7, 1): 0.016, (4096, 1, 2, 10, 1): 0.125, (1024, 1, 4, 7, 2): 0.016, (1024, 1, 5, 6, 4): 0.010, (1024, 1, 8, 4, 8): 0.005}
    for (N, k, l, beta, parties), m in measured.items():
        pred = tfhe.SchemeParameters(500, 1e-5, N, k, l, beta, 1e-9, 8, 2, 1e-5, parties).exactness()[2]
        assert 0.25 * pred < m <= pred, (N, k, l, beta, parties, m, pred)


def test_the_fuzzed_sets_outside_the_domain_are_class_0(tfhe):
    """profiles/r13_fuzz_params.txt (round 5): the sets whose MEASURED margin reached 1/4 are all class 0 by the a-priori law, and no
    set of class 1 or 2 in that log measured more than its prediction."""
    rows = []
    for line in open(os.path.join(ROOT, "profiles", "r13_fuzz_params.txt")):
        m = re.match(r"case\s+\d+ N=\s*(\d+) k=(\d+) l=\s*(\d+) beta=\s*(\d+) n=\s*\d+.*?margin ([0-9.]+)", line)
        if m:
            rows.append((int(m.group(1)), int(m.group(2)), int(m.group(3)), int(m.group(4)), float(m.group(5)), "skipped" in line))
    assert len(rows) >= 20 and sum(r[5] for r in rows) >= 8
    for N, k, l, beta, margin, skipped in rows:
        cls, bound, pred = tfhe.SchemeParameters(20, 1e-5, N, k, l, beta, 1e-9, 4, 2, 1e-5, 1).exactness()
        if skipped or margin >= 0.25:
            assert cls == 0, (N, k, l, beta, margin, pred)
        if cls >= 1:
            assert margin <= max(pred, 0.02), (N, k, l, beta, margin, pred)


@pytest.mark.gpu
def test_engine_states_the_same_class_and_the_constructor_warns_outside(tfhe):
    rng = np.random.default_rng(9)
    sets = [tfhe.tfhe_parameters_80(), tfhe.tfhe_parameters_128(), tfhe.tfhe_parameters_80(tlwe_mask_size=2), tfhe.mktfhe_parameters_2party,
            tfhe.mktfhe_parameters_8party, tfhe.SchemeParameters(630, 1 / 2**15, 2048, 1, 3, 7, 1 / 2**25, 8, 2, 1 / 2**15, 1)]
    for _ in range(40):
        N = 2 ** int(rng.integers(1, 14))
        beta = int(rng.integers(1, 13))
        l = int(rng.integers(1, 32 // beta + 1))
        parties = int(rng.choice([1, 1, 1, 2, 5]))
        k = 1 if parties > 1 else int(rng.integers(1, 7))
        sets.append(tfhe.SchemeParameters(8, 1e-5, N, k, l, beta, 1e-9, 4, 2, 1e-5, parties))
    seen = set()
    tfhe._lib._warned_inexact.clear()          # (earlier tests of the session may have met some of these sets)
    for p in sets:
        cls, bound, pred = p.exactness()
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            eng = tfhe.Engine(p)
            again = tfhe.Engine(p, devices=[0, 0])
        assert eng.get_option("exact_domain") == cls == again.get_option("exact_domain"), p
        assert abs(eng.get_option("exact_bound_log2_x1000") - round(bound * 1000)) <= 1
        assert abs(eng.get_option("exact_margin_x1e6") - pred * 1e6) <= 1 + 1e-6 * pred * 1e6
        key = p.engine_tuple()
        told = [x for x in w if issubclass(x.category, RuntimeWarning) and "exactness" in str(x.message)]
        assert len(told) == (1 if cls == 0 and key not in seen else 0), (p, cls, [str(x.message) for x in w])      # once per parameter set
        seen.add(key)
        eng.close(); again.close()
    assert {p.exactness()[0] for p in sets} == {0, 1, 2}
