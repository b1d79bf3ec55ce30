"""Checks of the bootstrap path that do NOT pass through oracle/tfhe_oracle.c (round-3 verdict, weak #1 / next #5).

Two independent anchors, both written here from the reference's text alone, in exact integer arithmetic:

(a) `Schoolbook`: the reference's gate-bootstrapping algebra restated with Python / numpy-int64 SCHOOLBOOK negacyclic
    products (np.convolve on int64: |digit| < 2^10, |key word| < 2^31, N <= 2048 terms -> every partial sum < 2^52, exact),
    no transform, no rounding, its own orchestration: decode_message (numeric-functions.jl:30-33), test vector and
    X^{-barb} (bootstrap.jl:50-56,78), CMUX acc + BK_i (.) ((X^a - 1) acc) (bootstrap.jl:19-23), gadget decomposition
    (tgsw.jl:99-117), the (p, j) x c product sum (tgsw.jl:125-129, tlwe.jl:105-111), extraction (tlwe.jl:55-59,
    polynomials.jl:32-35), keyswitch (keyswitch.jl:45-80).  Compared word for word with the oracle (CPU) and, under
    -m gpu, with tfhe_bootstrap_batch — the GPU tests call no oracle function.

(b) A noiseless known-answer test with a RANDOM mask: the bootstrapping key is a noiseless TGSW encryption of the LWE key
    bits (built here, not by the package's keygen), the input is b = <a, s> + phase with phase = +-1/4 exactly, so all n
    CMUX steps run with non-zero exponents.  What is left in the extracted phase besides +-mu is the gadget truncation:
    decompose() (tgsw.jl:104-116) adds the offset, shifts and masks, i.e. it FLOORS: sum_p digit_p g_p = c - (c mod h),
    h = 2^(32 - l beta), an error in (-h, 0] per coefficient with mean -(h - 1)/2, not zero.  One CMUX with key bit 1
    therefore changes the phase polynomial by e = eps_b - eps_a (*) K (K: the binary TLWE key, negacyclic product), whose
    coefficients have |mean| <= (h/2)(1 + |K|) and variance (h^2/12)(1 + |K|); later steps only rotate it.  Summed over the
    w = #{i: s_i = 1} steps:   |phase(extracted) -+ mu|  <=  w (h/2)(1 + |K|)  +  6 sqrt(w (h^2/12)(1 + |K|)).
    (The zero-mean formula sigma^2 = (n/2)(kN/2 + 1) 4^(31 - l beta)/3 misses the first term, which dominates.)  At the
    sizes used here the bound is 2 - 7 % of mu = 2^29; the worst case w h (1 + N) is useless at n = 500 (it exceeds mu).
"""
import numpy as np
import pytest

MASK32 = (1 << 32) - 1


def wrap32(v):
    """int64 array (or int) -> the same values as signed 32-bit words (wrapping)."""
    v = np.asarray(v, dtype=np.int64) & MASK32
    return np.where(v >= 2**31, v - 2**32, v).astype(np.int64)


def negacyclic(a, b, N):
    """a * b mod (X^N + 1) over the integers (int64 schoolbook; exact for the magnitudes used here)."""
    full = np.convolve(np.asarray(a, np.int64), np.asarray(b, np.int64))       # 2N - 1 coefficients
    out = full[:N].copy()
    out[: N - 1] -= full[N:]
    return out


def monomial(p, s, N):
    """X^s * p mod (X^N + 1), any integer s (DarkIntegers.mul_by_monomial as the reference calls it)."""
    s %= 2 * N
    out = np.empty(N, np.int64)
    for j in range(N):                     # coefficient j moves to j + s (mod 2N), sign flips past N
        d = (j + s) % (2 * N)
        if d < N:
            out[d] = p[j]
        else:
            out[d - N] = -p[j]
    return out


class Schoolbook:
    """Gate bootstrapping in exact integer arithmetic, orchestrated here (not in oracle/)."""

    def __init__(self, n, N, k, l, beta, t, gamma, bk, ks=None):
        self.n, self.N, self.k, self.l, self.beta, self.t, self.gamma = n, N, k, l, beta, t, gamma
        self.bk = np.asarray(bk, np.int64).reshape(n, l, k + 1, k + 1, N)      # key[i].samples[p, j].a[c]
        self.ks = None if ks is None else np.asarray(ks, np.int64).reshape(k * N, t, (1 << gamma) - 1, n + 1)

    def decode(self, phase, space):                                              # numeric-functions.jl:30-33
        log2 = space.bit_length() - 1
        v = int(wrap32(int(phase) + (1 << (32 - log2 - 1))))
        return v >> (32 - log2)                                                  # arithmetic shift: result in [-space/2, space/2)

    def decompose(self, poly):                                                   # tgsw.jl:99-117
        l, beta = self.l, self.beta
        offset = sum(1 << (32 - p * beta + beta - 1) for p in range(1, l + 1))
        c = wrap32(np.asarray(poly, np.int64) + offset)
        half, mask = 1 << (beta - 1), (1 << beta) - 1
        return [((c >> (32 - p * beta)) & mask) - half for p in range(1, l + 1)]

    def extern_mul(self, temp, i):                                               # tgsw.jl:125-129
        k, N = self.k, self.N
        out = [np.zeros(N, np.int64) for _ in range(k + 1)]
        for j in range(k + 1):
            for p, d in enumerate(self.decompose(temp[j])):
                for c in range(k + 1):
                    out[c] += negacyclic(d, self.bk[i, p, j, c], N)
        return [wrap32(o) for o in out]

    def bootstrap_wo_keyswitch(self, mu, x):                                     # bootstrap.jl:69-82
        n, N, k = self.n, self.N, self.k
        bara = [self.decode(v, 2 * N) for v in x[:n]]
        barb = self.decode(x[n], 2 * N)
        acc = [np.zeros(N, np.int64) for _ in range(k)] + [monomial(np.full(N, mu, np.int64), -barb, N)]
        for i in range(n):                                                       # bootstrap.jl:32-39
            if bara[i] == 0:
                continue
            temp = [wrap32(monomial(a, bara[i], N) - a) for a in acc]            # bootstrap.jl:21
            prod = self.extern_mul(temp, i)
            acc = [wrap32(a + q) for a, q in zip(acc, prod)]                     # bootstrap.jl:22
        ext = np.empty(k * N + 1, np.int64)                                      # tlwe.jl:55-59: reverse, then X^(N+1)... = a'[0] = p[0], a'[m] = -p[N-m]
        for c in range(k):
            ext[c * N] = acc[c][0]
            ext[c * N + 1: (c + 1) * N] = -acc[c][:0:-1]
        ext[k * N] = acc[k][0]
        return wrap32(ext)

    def keyswitch(self, u):                                                      # keyswitch.jl:45-80
        n, kN, t, gamma = self.n, self.k * self.N, self.t, self.gamma
        res = np.zeros(n + 1, np.int64)
        res[n] = u[kN]
        abar = wrap32(np.asarray(u[:kN], np.int64) + (1 << (32 - (1 + gamma * t))))
        for i in range(kN):
            for j in range(1, t + 1):
                d = (int(abar[i]) >> (32 - j * gamma)) & ((1 << gamma) - 1)
                if d:
                    res -= self.ks[i, j - 1, d - 1]
        return wrap32(res)

    def bootstrap(self, mu, x):                                                  # bootstrap.jl:92-95
        return self.keyswitch(self.bootstrap_wo_keyswitch(mu, x))


def _inputs(rng, rows, n):
    x = rng.integers(-2**31, 2**31, size=(rows, n + 1), dtype=np.int64).astype(np.int32)
    x[0, :] = 0                       # every exponent zero: the branch at bootstrap.jl:34
    if n >= 2:
        x[1, :2] = [2**31 - 1, -2**31]
    return x


CASES = [  # N, k, l, beta, n
    (1024, 1, 2, 10, 3),              # tfhe_parameters_80's TLWE / gadget shape (api.jl:30-52)
    (1024, 1, 3, 7, 2),               # tfhe_parameters_128's (api.jl:55-69)
    (1024, 2, 2, 10, 2),              # tlwe_mask_size = 2 (api.jl:30 keyword)
    (2048, 1, 3, 7, 2),               # BASELINE config 4b's synthetic N = 2048 shape
    (512, 1, 2, 10, 3),               # any power-of-two N (api.jl:4-21 validates nothing): the any-N kernel (round 5)
    (4096, 1, 3, 7, 2),
    (256, 2, 2, 8, 2),
    (1024, 5, 1, 8, 2),               # tlwe_mask_size beyond what blind_rotate_kernel_general holds
]


def _keys(tfhe, N, k, l, beta, n, seed):
    p = tfhe.SchemeParameters(n, 1 / 2**15, N, k, l, beta, 9e-9, 8, 2, 1 / 2**15, 1)
    rng = np.random.default_rng(seed)
    sk, ck = tfhe.make_key_pair(rng, p)
    return p, rng, sk, ck


@pytest.mark.parametrize("N,k,l,beta,n", CASES)
def test_oracle_equals_schoolbook(orc, tfhe, N, k, l, beta, n):
    """The C oracle (both of its product back-ends) against the integer schoolbook restatement above."""
    p, rng, sk, ck = _keys(tfhe, N, k, l, beta, n, 900 + N + k + l)
    sb = Schoolbook(n, N, k, l, beta, 8, 2, ck.bootstrap_key, ck.keyswitch_key)
    o = orc.Oracle(n, N, k, l, beta, 8, 2)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    x = _inputs(rng, 3, n)
    want_ext = np.stack([sb.bootstrap_wo_keyswitch(2**29, row) for row in x])
    want = np.stack([sb.keyswitch(e) for e in want_ext])
    for mode in (orc.MODE_FFT, orc.MODE_EXACT):
        assert np.array_equal(o.bootstrap(2**29, x, with_keyswitch=False, mode=mode), want_ext.astype(np.int32))
    assert np.array_equal(o.bootstrap(2**29, x, with_keyswitch=True), want.astype(np.int32))
    ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,k,l,beta,n", CASES)
def test_gpu_equals_schoolbook(tfhe, N, k, l, beta, n):
    """tfhe_bootstrap_batch (with and without keyswitch) against the schoolbook restatement — no oracle call.  Small batches
    take the multi-wave kernels; the one-wave kernels (N = 1024, k = 1) are forced with br_tiny / br_small."""
    p, rng, sk, ck = _keys(tfhe, N, k, l, beta, n, 900 + N + k + l)
    sb = Schoolbook(n, N, k, l, beta, 8, 2, ck.bootstrap_key, ck.keyswitch_key)
    x = _inputs(rng, 3, n)
    want_ext = np.stack([sb.bootstrap_wo_keyswitch(2**29, row) for row in x]).astype(np.int32)
    want = np.stack([sb.keyswitch(e) for e in want_ext]).astype(np.int32)
    eng = ck.engine(0)
    seen = set()
    settings = [{}] + ([{"br_tiny": -1}, {"br_tiny": -1, "br_small": -1}, {"br_tiny": -1, "br_small": -1, "v3_rw": 4}] if (N, k) == (1024, 1) else [])
    for opts in settings:
        for name, v in opts.items():
            eng.set_option(name, v)
        assert np.array_equal(eng.bootstrap(2**29, x, with_keyswitch=False), want_ext), opts
        seen.add(eng.last_kernel_name())
        assert np.array_equal(eng.bootstrap(2**29, x, with_keyswitch=True), want), opts
    assert len(seen) == len(settings), seen
    ck.close()


# ---- (a-gates) every gate's affine prologue + MUX's two rotations, restated from gates.jl ----------------------------------------
# result = c/8-or-/4 constant on b  +  sx * x  +  sy * y  (then one bootstrap with mu = 1/8); gates.jl:15-18 (NAND), :27-30 (OR),
# :39-42 (AND), :51-54 (XOR: (x + y) * 2, every word wraps), :63-66 (XNOR), :100-103 (NOR), :113-116 (ANDNY), :126-129 (ANDYN),
# :139-142 (ORNY), :152-155 (ORYN)
_AFFINE = {"NAND": (2**29, -1, -1), "OR": (2**29, 1, 1), "AND": (-2**29, 1, 1), "XOR": (2**30, 2, 2), "XNOR": (-2**30, -2, -2),
           "NOR": (-2**29, -1, -1), "ANDNY": (-2**29, -1, 1), "ANDYN": (-2**29, 1, -1), "ORNY": (2**29, -1, 1), "ORYN": (2**29, 1, -1)}


def schoolbook_gate(sb, name, x, y, z):
    """One gate of gates.jl on int32 sample rows, every word wrapping (lwe.jl:63-82)."""
    x, y, z = (np.asarray(v, np.int64) for v in (x, y, z))
    n = sb.n
    if name in _AFFINE:
        const, sx, sy = _AFFINE[name]
        t = wrap32(sx * x + sy * y)
        t[n] = wrap32(t[n] + const)
        return sb.bootstrap(2**29, t)
    if name == "NOT":                                        # gates.jl:76-79: not bootstrapped
        return wrap32(-x)
    if name == "COPY":                                       # (this engine's own opcode: the sample unchanged)
        return wrap32(x)
    if name in ("CONST0", "CONST1"):                         # gates.jl:91-93: noiseless trivial sample of -1/8 / +1/8
        r = np.zeros(n + 1, np.int64)
        r[n] = 2**29 if name == "CONST1" else -2**29
        return wrap32(r)
    assert name == "MUX"                                     # gates.jl:163-177
    t1 = wrap32(x + y); t1[n] = wrap32(t1[n] - 2**29)        # AND(x, y)
    t2 = wrap32(-x + z); t2[n] = wrap32(t2[n] - 2**29)       # AND(NOT x, z)
    u1, u2 = sb.bootstrap_wo_keyswitch(2**29, t1), sb.bootstrap_wo_keyswitch(2**29, t2)
    t3 = wrap32(u1 + u2); t3[-1] = wrap32(t3[-1] + 2**29)    # OR in the extracted dimension, then ONE keyswitch
    return sb.keyswitch(t3)


@pytest.mark.parametrize("N,k,l,beta,n", CASES[:3])
def test_oracle_gates_equal_schoolbook_gates(orc, tfhe, N, k, l, beta, n):
    """The oracle's gate layer (all 15 opcodes, arbitrary input words) against gates.jl restated above."""
    p, rng, sk, ck = _keys(tfhe, N, k, l, beta, n, 1900 + N + k + l)
    sb = Schoolbook(n, N, k, l, beta, 8, 2, ck.bootstrap_key, ck.keyswitch_key)
    o = orc.Oracle(n, N, k, l, beta, 8, 2)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    order = list(tfhe.OPCODES) + ["MUX", "XOR", "NAND"]
    B = len(order)
    x, y, z = (rng.integers(-2**31, 2**31, size=(B, n + 1), dtype=np.int64).astype(np.int32) for _ in range(3))
    ops = np.array([tfhe.OPCODES[nm] for nm in order], np.uint8)
    got = o.gates(ops, x, y, z)
    for g, nm in enumerate(order):
        assert np.array_equal(got[g], schoolbook_gate(sb, nm, x[g], y[g], z[g]).astype(np.int32)), nm
    ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,k,l,beta,n", CASES[:3])
def test_gpu_gates_equal_schoolbook_gates(tfhe, N, k, l, beta, n):
    """BASELINE config 3's gate mix and every other opcode through tfhe_gates_batch against gates.jl restated above on top of the
    schoolbook bootstrap — arbitrary input words, no oracle call."""
    p, rng, sk, ck = _keys(tfhe, N, k, l, beta, n, 1900 + N + k + l)
    sb = Schoolbook(n, N, k, l, beta, 8, 2, ck.bootstrap_key, ck.keyswitch_key)
    names = list(tfhe.OPCODES)
    order = names + ["MUX", "XOR", "NAND"]
    B = len(order)
    x, y, z = (rng.integers(-2**31, 2**31, size=(B, n + 1), dtype=np.int64).astype(np.int32) for _ in range(3))
    ops = np.array([tfhe.OPCODES[nm] for nm in order], np.uint8)
    want = np.stack([schoolbook_gate(sb, nm, x[g], y[g], z[g]) for g, nm in enumerate(order)]).astype(np.int32)
    got = ck.engine(0).gates(ops, x, y, z)
    for g, nm in enumerate(order):
        assert np.array_equal(got[g], want[g]), nm
    ck.close()


# ---- (b) noiseless KAT with a random mask ---------------------------------------------------------------------------
def _noiseless_setup(n, N, l, beta, seed):
    """LWE key, binary TLWE key, noiseless TGSW(s_i) rows built here: row (p, j) = (a, a (*) K) + s_i g_p on component j
    (tgsw.jl:52-72 with zero noise), as the canonical Int32 key [n][l][2][2][N]."""
    rng = np.random.default_rng(seed)
    s = rng.integers(0, 2, n).astype(np.int64)
    K = rng.integers(0, 2, N).astype(np.int64)
    bk = np.zeros((n, l, 2, 2, N), np.int64)
    for i in range(n):
        for p in range(l):
            for j in range(2):
                a = rng.integers(-2**31, 2**31, N, dtype=np.int64)
                bk[i, p, j, 0] = a
                bk[i, p, j, 1] = wrap32(negacyclic(a, K, N))
                bk[i, p, j, j, 0] = wrap32(bk[i, p, j, j, 0] + s[i] * (1 << (32 - (p + 1) * beta)))
    return rng, s, K, bk.astype(np.int32)


def _kat_inputs(rng, s, n, rows):
    """b = <a, s> + phase, phase = +1/4 (even rows) / -1/4 (odd rows), random masks, no noise."""
    a = rng.integers(-2**31, 2**31, size=(rows, n), dtype=np.int64)
    phase = np.where(np.arange(rows) % 2 == 0, 2**30, -2**30)
    b = wrap32(a @ s + phase)
    return np.concatenate([a, b[:, None]], axis=1).astype(np.int32), phase


def _kat_check(ext, phase, s, K, N, l, beta, mu):
    h = 1 << (32 - l * beta)
    w, ones = int(s.sum()), int(K.sum())
    bound = w * (h / 2) * (1 + ones) + 6 * np.sqrt(w * (h * h / 12) * (1 + ones))
    assert bound < mu / 8                                       # the check means something
    ph = wrap32(ext[:, N].astype(np.int64) - ext[:, :N].astype(np.int64) @ K)       # extracted LWE key = coefficients of K
    err = ph - np.where(phase > 0, mu, -mu)
    assert np.abs(err).max() <= bound, (np.abs(err).max(), bound)
    assert np.abs(err).max() > 0                                # the truncation error is really there (not a trivial pass)
    return float(np.abs(err).max() / bound)


KAT = [(32, 1024, 2, 10), (24, 1024, 3, 7), (16, 2048, 3, 7)]


@pytest.mark.parametrize("n,N,l,beta", KAT)
def test_noiseless_random_mask_kat_oracle(orc, n, N, l, beta):
    rng, s, K, bk = _noiseless_setup(n, N, l, beta, 5 + n)
    x, phase = _kat_inputs(rng, s, n, 6)
    assert (np.abs(wrap32((x[:, :n].astype(np.int64) + 2**20) >> 21)) > 0).mean() > 0.9     # the exponents are not zero
    o = orc.Oracle(n, N, 1, l, beta, 8, 2)
    o.load_bootstrap_key(bk)
    _kat_check(o.bootstrap(2**29, x, with_keyswitch=False), phase, s, K, N, l, beta, 2**29)


@pytest.mark.gpu
@pytest.mark.parametrize("n,N,l,beta", KAT)
def test_noiseless_random_mask_kat_gpu(tfhe, n, N, l, beta):
    """The same known answer from tfhe_bootstrap_batch alone: no oracle, no package keygen."""
    rng, s, K, bk = _noiseless_setup(n, N, l, beta, 5 + n)
    x, phase = _kat_inputs(rng, s, n, 6)
    eng = tfhe._lib.Engine(tfhe.SchemeParameters(n, 1 / 2**15, N, 1, l, beta, 0.0, 8, 2, 1 / 2**15, 1))
    eng.load_bootstrap_key(bk)
    _kat_check(eng.bootstrap(2**29, x, with_keyswitch=False), phase, s, K, N, l, beta, 2**29)
    if N == 1024:
        eng.set_option("br_tiny", -1); eng.set_option("br_small", -1)           # the one-wave kernel as well
        _kat_check(eng.bootstrap(2**29, x, with_keyswitch=False), phase, s, K, N, l, beta, 2**29)
    eng.close()


# ---- (a') the schoolbook restatement at the SHIPPED sizes --------------------------------------------------------------------
def _full_size_case(tfhe, keys, rows):
    K = keys
    p = K.params
    sb = Schoolbook(p.lwe_size, p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length, p.bs_log2_base,
                    p.ks_decomp_length, p.ks_log2_base, K.ck.bootstrap_key, K.ck.keyswitch_key)
    rng = np.random.default_rng(8080)
    bits = rng.integers(0, 2, (2, rows)).astype(bool)
    x, y = tfhe.encrypt(K.rng, K.sk, bits[0]).data, tfhe.encrypt(K.rng, K.sk, bits[1]).data
    # the NAND prologue written here (gates.jl:15-18: (0, 1/8) - x - y), then the schoolbook bootstrap with mu = 1/8
    pre = wrap32(-x.astype(np.int64) - y.astype(np.int64))
    pre[:, -1] = wrap32(pre[:, -1] + 2**29)
    want = np.stack([sb.bootstrap(2**29, row) for row in pre]).astype(np.int32)
    return x, y, bits, want


def test_oracle_equals_schoolbook_full_size_80bit(orc, tfhe, keys80):
    """tfhe_parameters_80 at its shipped size (n = 500, 500 CMUX steps): one NAND gate of the oracle against the integer
    schoolbook restatement, word for word (4 s of numpy convolutions)."""
    x, y, bits, want = _full_size_case(tfhe, keys80, 1)
    got = keys80.oracle.gates(np.zeros(1, np.uint8), x, y)
    assert np.array_equal(got, want)
    assert list(tfhe.decrypt(keys80.sk, got)) == [not (bits[0][0] and bits[1][0])]


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["80", "128"])
def test_gpu_equals_schoolbook_full_size(tfhe, keys80, keys128, which):
    """The shipped parameter sets at full size: NAND gates through tfhe_gates_batch against the schoolbook restatement —
    no oracle call.  Two gates each (500 / 630 CMUX steps of eight / twelve 1024-coefficient integer products)."""
    K = keys80 if which == "80" else keys128
    x, y, bits, want = _full_size_case(tfhe, K, 2)
    got = K.ck.engine(0).gates(np.zeros(2, np.uint8), x, y)
    assert np.array_equal(got, want)
    assert np.array_equal(tfhe.decrypt(K.sk, got), ~(bits[0] & bits[1]))


@pytest.mark.gpu
def test_gpu_equals_schoolbook_full_size_n2048(tfhe, orc):
    """BASELINE config 4b at full size (synthetic N = 2048, n = 630, l = 3, beta = 7): one NAND gate through tfhe_gates_batch
    (blind_rotate_kernel_n2048x) against the schoolbook restatement — no oracle call in the comparison (630 CMUX steps of twelve
    2048-coefficient integer products: about a minute of numpy convolutions)."""
    from conftest import KeySet
    from test_oracle import synthetic_2048
    K = KeySet(tfhe, orc, synthetic_2048(tfhe), seed=2048)
    x, y, bits, want = _full_size_case(tfhe, K, 1)
    eng = K.ck.engine(0)
    got = eng.gates(np.zeros(1, np.uint8), x, y)
    assert eng.last_kernel_name().startswith("blind_rotate_kernel_n2048x<3")
    assert np.array_equal(got, want)
    assert np.array_equal(tfhe.decrypt(K.sk, got), ~(bits[0] & bits[1]))
    K.ck.close()


# ---- (a'') multi-key (config 5): mk_gate_nand in exact integer arithmetic ------------------------------------------------------
class MKSchoolbook(Schoolbook):
    """mk_gate_nand (mk_gates.jl:7-12) -> mk_bootstrap (mk_internals.jl:464-515) -> mk_tgsw_extern_mul (:348-391) ->
    mk_keyswitch (:397-411), restated with integer schoolbook products.  Key: Int32 [P][n][2 l P + 2 l][N], per (party i,
    bit j) the polynomials x[l][P] | y[l][P] | c0[l] | c1[l] (include/tfhe_mi355x.h); samples: a[:, 1]; ...; a[:, P]; b."""

    def __init__(self, n, N, l, beta, t, gamma, parties, bk, ks):
        self.n, self.N, self.k, self.l, self.beta, self.t, self.gamma, self.P = n, N, 1, l, beta, t, gamma, parties
        self.mbk = np.asarray(bk, np.int64).reshape(parties, n, 2 * l * parties + 2 * l, N)
        self.mks = np.asarray(ks, np.int64).reshape(parties, N, t, (1 << gamma) - 1, n + 1)

    def mk_extern_mul(self, temp, key, party):                                   # mk_internals.jl:348-391
        P, l, N = self.P, self.l, self.N
        x = lambda p, q: key[p * P + q]
        y = lambda p, q: key[l * P + p * P + q]
        c0 = lambda p: key[2 * l * P + p]
        c1 = lambda p: key[2 * l * P + l + p]
        da = [self.decompose(temp[i]) for i in range(P)]                         # da[i][p]
        db = self.decompose(temp[P])
        out = []
        for i in range(P):
            if i == party:                                                       # c'_party = sum g^-1(a_j) y_j + g^-1(b) c1
                acc = sum(negacyclic(da[j][p], y(p, j), N) for p in range(l) for j in range(P))
                acc = acc + sum(negacyclic(db[p], c1(p), N) for p in range(l))
            else:                                                                # c'_i = g^-1(a_i) y_party
                acc = sum(negacyclic(da[i][p], y(p, party), N) for p in range(l))
            out.append(wrap32(acc))
        body = sum(negacyclic(da[i][p], x(p, i), N) for p in range(l) for i in range(P))
        body = body + sum(negacyclic(db[p], c0(p), N) for p in range(l))
        out.append(wrap32(body))
        return out

    def mk_gate_nand(self, xs, ys):
        n, N, P = self.n, self.N, self.P
        temp = wrap32(-xs.astype(np.int64) - ys.astype(np.int64))               # mk_gates.jl:8-11
        temp[P * n] = int(wrap32(temp[P * n] + 2**29))
        bara = [[self.decode(temp[i * n + j], 2 * N) for j in range(n)] for i in range(P)]
        barb = self.decode(temp[P * n], 2 * N)
        acc = [np.zeros(N, np.int64) for _ in range(P)] + [monomial(np.full(N, 2**29, np.int64), -barb, N)]
        for i in range(P):                                                       # party-major: mk_internals.jl:475-476
            for j in range(n):
                if bara[i][j] == 0:
                    continue
                rot = [wrap32(monomial(a, bara[i][j], N) - a) for a in acc]      # mk_mux_rotate :464-470
                prod = self.mk_extern_mul(rot, self.mbk[i, j], i)
                acc = [wrap32(a + q) for a, q in zip(acc, prod)]
        res = np.zeros(P * n + 1, np.int64)
        res[P * n] = acc[P][0]                                                   # mk_tlwe_extract_sample :88-95, then mk_keyswitch
        for p in range(P):
            u = np.empty(N + 1, np.int64)
            u[0] = acc[p][0]
            u[1:N] = -acc[p][:0:-1]
            u[N] = 0                                                             # b = 0 per party (:399-401)
            self.ks, self.k = self.mks[p], 1
            part = self.keyswitch(wrap32(u))
            res[p * n:(p + 1) * n] = part[:n]
            res[P * n] += part[n]
        return wrap32(res)


def _mk_setup(tfhe, parties, l, beta, n, seed, N=1024, t=8, gamma=2):
    p = tfhe.SchemeParameters(n, 0.012467, N, 1, l, beta, 3.29e-10, t, gamma, 2.44e-5, parties)
    rng = np.random.default_rng(seed)
    sks = [tfhe.SecretKey(rng, p) for _ in range(parties)]
    shared = tfhe.SharedKey(rng, p)
    ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, sk, shared) for sk in sks])
    sb = MKSchoolbook(n, N, l, beta, t, gamma, parties, ck.bootstrap_key, ck.keyswitch_key)
    xs = tfhe.mk_encrypt(rng, sks, [True, False, True])
    ys = tfhe.mk_encrypt(rng, sks, [True, True, False])
    extra = rng.integers(-2**31, 2**31, size=(1, parties * n + 1), dtype=np.int64).astype(np.int32)     # an arbitrary word row
    xs, ys = np.concatenate([xs, extra]), np.concatenate([ys, extra[:, ::-1]])
    want = np.stack([sb.mk_gate_nand(a, b) for a, b in zip(xs, ys)]).astype(np.int32)
    return p, sks, ck, xs, ys, want


MK_CASES = [(2, 4, 7, 6), (4, 5, 6, 3), (8, 8, 4, 2)]      # (parties, l, beta, n): mktfhe_parameters_2party / _4party / _8party gadget shapes (mk_api.jl:4-34)


@pytest.mark.parametrize("parties,l,beta,n", MK_CASES)
def test_oracle_mk_equals_schoolbook(orc, tfhe, parties, l, beta, n):
    p, sks, ck, xs, ys, want = _mk_setup(tfhe, parties, l, beta, n, 400 + parties)
    o = orc.Oracle(n, 1024, 1, l, beta, 8, 2, parties=parties)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    assert np.array_equal(o.mk_gate_nand(xs, ys, nthreads=4), want)
    assert list(tfhe.mk_decrypt(sks, want[:3])) == [False, True, True]
    ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize("parties,l,beta,n", MK_CASES)
def test_gpu_mk_equals_schoolbook(tfhe, parties, l, beta, n):
    """tfhe_mk_gate_nand_batch (mk_blind_rotate_kernel_w2<4> / g2<4,5,acc=lds> / g2<8,8>) against the multi-key schoolbook — no oracle."""
    p, sks, ck, xs, ys, want = _mk_setup(tfhe, parties, l, beta, n, 400 + parties)
    eng = ck.engine(0)
    assert np.array_equal(eng.mk_gate_nand(xs, ys), want)
    assert eng.last_kernel_name() == {2: "mk_blind_rotate_kernel_w2<4>", 4: "mk_blind_rotate_kernel_g2<4,5,acc=lds>", 8: "mk_blind_rotate_kernel_g2<8,8>"}[parties]
    ck.close()


@pytest.mark.gpu
def test_gpu_mk_equals_schoolbook_full_size_2party(tfhe):
    """BASELINE config 5 at full size (mktfhe_parameters_2party, mk_api.jl:4-14): one 2-party NAND through
    tfhe_mk_gate_nand_batch (mk_blind_rotate_kernel_w2<4>) against the multi-key schoolbook — no oracle (2 x n CMUX steps of
    28 integer products: under a minute of numpy convolutions)."""
    p = tfhe.mktfhe_parameters_2party
    rng = np.random.default_rng(52)
    sks = [tfhe.SecretKey(rng, p) for _ in range(2)]
    shared = tfhe.SharedKey(rng, p)
    ck = tfhe.MKCloudKey([tfhe.CloudKeyPart(rng, sk, shared) for sk in sks])
    sb = MKSchoolbook(p.lwe_size, p.tlwe_polynomial_degree, p.bs_decomp_length, p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base,
                      2, ck.bootstrap_key, ck.keyswitch_key)
    xs, ys = tfhe.mk_encrypt(rng, sks, [True]), tfhe.mk_encrypt(rng, sks, [True])
    want = sb.mk_gate_nand(xs[0], ys[0]).astype(np.int32)
    eng = ck.engine(0)
    got = eng.mk_gate_nand(xs, ys)
    assert eng.last_kernel_name() == "mk_blind_rotate_kernel_w2<4>"
    assert np.array_equal(got[0], want)
    ck.close()



# ---- multi-key sets outside the shipped shapes (round 5: the engine refuses none of them) -----------------------------------------
MK_ANY_CASES = [  # parties, l, beta, n, N, t, gamma, kernel
    (2, 4, 7, 3, 1024, 5, 3, "mk_blind_rotate_kernel_w2<4>"),                 # keyswitch t = 5 / base 8 (mk_internals.jl:397-411 takes any)
    (2, 4, 7, 3, 512, 8, 2, "mk_blind_rotate_kernel_anyn(N=512,P=2,l=4)"),    # another polynomial degree
    (9, 8, 4, 2, 1024, 8, 2, "mk_blind_rotate_kernel_anyn(N=1024,P=9,l=8)"),  # more parties than any shipped set
]


@pytest.mark.parametrize("parties,l,beta,n,N,t,gamma,kernel", MK_ANY_CASES)
def test_oracle_mk_any_equals_schoolbook(orc, tfhe, parties, l, beta, n, N, t, gamma, kernel):
    p, sks, ck, xs, ys, want = _mk_setup(tfhe, parties, l, beta, n, 700 + parties + N + t, N=N, t=t, gamma=gamma)
    o = orc.Oracle(n, N, 1, l, beta, t, gamma, parties=parties)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    assert np.array_equal(o.mk_gate_nand(xs, ys, nthreads=4), want)
    ck.close()


@pytest.mark.gpu
@pytest.mark.parametrize("parties,l,beta,n,N,t,gamma,kernel", MK_ANY_CASES)
def test_gpu_mk_any_equals_schoolbook(tfhe, parties, l, beta, n, N, t, gamma, kernel):
    """tfhe_mk_gate_nand_batch on multi-key sets outside the shipped shapes against the multi-key schoolbook — no oracle."""
    p, sks, ck, xs, ys, want = _mk_setup(tfhe, parties, l, beta, n, 700 + parties + N + t, N=N, t=t, gamma=gamma)
    eng = ck.engine(0)
    assert np.array_equal(eng.mk_gate_nand(xs, ys), want)
    assert eng.last_kernel_name() == kernel, eng.last_kernel_name()
    ck.close()
