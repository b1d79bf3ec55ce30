"""ctypes binding of oracle/libtfhe_oracle.so — the CPU restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see the header of oracle/tfhe_oracle.c): imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product package.
Ciphertext-word parity is "parity unpinned" by the reference's own tests (it ships no Int32
fixtures); this oracle is anchored on the reference's truth tables, closed-form constants and the
agreement of its two product back-ends (reference-style Float64 FFT vs exact integer).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TFHE_ORACLE_SO: another build of the same source (the AddressSanitizer / UBSan build of `make -C oracle sanitize`,
# tests/test_oracle_sanitized.py)
_SO = os.environ.get("TFHE_ORACLE_SO") or os.path.join(_HERE, "libtfhe_oracle.so")

# opcode numbering shared with include/tfhe_mi355x.h
OPS = dict(NAND=0, OR=1, AND=2, XOR=3, XNOR=4, NOT=5, NOR=6, ANDNY=7, ANDYN=8, ORNY=9, ORYN=10,
           MUX=11, CONST0=12, CONST1=13, COPY=14)

MODE_FFT = 0     # the reference's folded Float64 FFT + round (polynomials.jl:106-132)
MODE_EXACT = 1   # exact negacyclic product mod 2^32


class OrcParams(C.Structure):
    _fields_ = [(f, C.c_int32) for f in ("n", "N", "k", "l", "log2Bg", "t", "log2ks", "parties")]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "tfhe_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_encode_message.restype = C.c_int32
        _lib.orc_decode_message.restype = C.c_int32
        _lib.orc_max_threads.restype = C.c_int32
    return _lib


def _p(a, ty=C.c_void_p):
    if a is None:
        return None
    return a.ctypes.data_as(ty)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def encode_message(mu, space):
    return int(lib().orc_encode_message(C.c_int32(mu), C.c_int32(int(space).bit_length() - 1)))


def decode_message(phase, space):
    return int(lib().orc_decode_message(C.c_int32(phase), C.c_int32(int(space).bit_length() - 1)))


def tgsw_constants(l, log2Bg):
    g = np.zeros(l, np.int32)
    off = C.c_int32(0)
    lib().orc_tgsw_constants(C.c_int32(l), C.c_int32(log2Bg), _p(g), C.byref(off))
    return g, off.value


def decompose(poly, l, log2Bg):
    poly = _i32(poly)
    out = np.zeros((l, poly.size), np.int32)
    lib().orc_decompose(_p(poly), C.c_int32(poly.size), C.c_int32(l), C.c_int32(log2Bg), _p(out))
    return out


def mul_by_monomial(poly, s):
    poly = _i32(poly)
    out = np.zeros_like(poly)
    lib().orc_mul_by_monomial(_p(poly), C.c_int32(poly.size), C.c_int32(int(s)), _p(out))
    return out


def negacyclic_mul_exact(a, b):
    a, b = _i32(a), _i32(b)
    out = np.zeros_like(a)
    lib().orc_negacyclic_mul_exact(_p(a), _p(b), C.c_int32(a.size), _p(out))
    return out


def negacyclic_mul_fft(a, b):
    """Returns (product, worst |pre-round value - nearest integer|)."""
    a, b = _i32(a), _i32(b)
    assert lib().orc_init(C.c_int32(a.size)) == 0
    out = np.zeros_like(a)
    m = C.c_double(0)
    assert lib().orc_negacyclic_mul_fft(_p(a), _p(b), C.c_int32(a.size), _p(out), C.byref(m)) == 0
    return out, m.value


def forward_transform(poly):
    poly = _i32(poly)
    assert lib().orc_init(C.c_int32(poly.size)) == 0
    re = np.zeros(poly.size // 2)
    im = np.zeros(poly.size // 2)
    assert lib().orc_forward_transform(_p(poly), C.c_int32(poly.size), _p(re), _p(im)) == 0
    return re + 1j * im


def max_threads():
    return int(lib().orc_max_threads())


class Oracle:
    """Holds one cloud key (bootstrapping key as Int32 and as reference-style spectra, keyswitch key)."""

    def __init__(self, n, N, k, l, log2Bg, t, log2ks, parties=1):
        self.P = OrcParams(n, N, k, l, log2Bg, t, log2ks, parties)
        self.n, self.N, self.k, self.l, self.parties = n, N, k, l, parties
        assert lib().orc_init(C.c_int32(N)) == 0
        self.bk_i32 = self.bk_re = self.bk_im = self.ks = None

    # bk: Int32, any shape whose last axis is N (single key [n][l][k+1][k+1][N]; MK see tfhe_oracle.c)
    def load_bootstrap_key(self, bk_i32):
        self.bk_i32 = _i32(bk_i32)
        npolys = self.bk_i32.size // self.N
        self.bk_re = np.zeros(npolys * (self.N // 2))
        self.bk_im = np.zeros(npolys * (self.N // 2))
        rc = lib().orc_bk_transform(C.byref(self.P), _p(self.bk_i32), _p(self.bk_re), _p(self.bk_im),
                                    C.c_int64(npolys))
        assert rc == 0

    def load_bootstrap_spectra(self, spectra):
        """The key in the reference's stored form (complex128 [..., N/2], bootstrap.jl:12-14 / mk_internals.jl:442-461),
        e.g. from a fixture minted by julia/TFHEMI355X/scripts/mint_fixtures.jl: the FFT back-end then multiplies with exactly the
        reference's spectra; the Int32 form (for the exact back-end) is their inverse transform (polynomials.jl:119-132)."""
        sp = np.ascontiguousarray(spectra, dtype=np.complex128)
        M = self.N // 2
        assert sp.shape[-1] == M
        flat = sp.reshape(-1, M)
        self.bk_re = np.ascontiguousarray(flat.real).reshape(-1)
        self.bk_im = np.ascontiguousarray(flat.imag).reshape(-1)
        out = np.zeros((flat.shape[0], self.N), np.int32)
        worst = 0.0
        for q in range(flat.shape[0]):
            re, im = np.ascontiguousarray(flat[q].real), np.ascontiguousarray(flat[q].imag)
            m = C.c_double(0)
            assert lib().orc_inverse_transform(_p(re), _p(im), C.c_int32(self.N), _p(out[q]), C.byref(m)) == 0
            worst = max(worst, m.value)
        assert worst < 0.25, f"spectra are not those of integer polynomials (margin {worst})"
        self.bk_i32 = out.reshape(sp.shape[:-1] + (self.N,))

    def load_keyswitch_key(self, ks):
        self.ks = _i32(ks)

    def bk_spectra(self):
        """The reference's stored form (bootstrap.jl:12-14): complex128 [..., N/2]."""
        return (self.bk_re + 1j * self.bk_im).reshape(self.bk_i32.shape[:-1] + (self.N // 2,))

    def gates(self, ops, in0, in1=None, in2=None, mode=MODE_FFT, nthreads=0):
        ops = np.ascontiguousarray(ops, dtype=np.uint8)
        in0 = _i32(in0)
        B = ops.size
        n1 = self.n + 1
        assert in0.shape == (B, n1)
        in1 = _i32(in1) if in1 is not None else np.zeros_like(in0)
        in2 = _i32(in2) if in2 is not None else np.zeros_like(in0)
        out = np.zeros((B, n1), np.int32)
        margin = C.c_double(0)
        rc = lib().orc_gates_batch(C.byref(self.P), _p(self.bk_re), _p(self.bk_im), _p(self.bk_i32),
                                   _p(self.ks), C.c_int32(mode), _p(ops), _p(in0), _p(in1), _p(in2),
                                   _p(out), C.c_int64(B), C.c_int32(nthreads), C.byref(margin))
        assert rc == 0, "oracle gate failed"
        self.last_margin = margin.value
        return out

    def bootstrap(self, mu, x, with_keyswitch=True, mode=MODE_FFT, nthreads=0):
        x = _i32(x)
        B = x.shape[0]
        width = self.n + 1 if with_keyswitch else self.k * self.N + 1
        out = np.zeros((B, width), np.int32)
        rc = lib().orc_bootstrap_batch(C.byref(self.P), _p(self.bk_re), _p(self.bk_im), _p(self.bk_i32),
                                       _p(self.ks), C.c_int32(mode), C.c_int32(mu), _p(x), _p(out),
                                       C.c_int64(B), C.c_int32(1 if with_keyswitch else 0),
                                       C.c_int32(nthreads))
        assert rc == 0
        return out

    def keyswitch(self, x):
        x = _i32(x)
        B = x.shape[0]
        out = np.zeros((B, self.n + 1), np.int32)
        assert lib().orc_keyswitch_batch(C.byref(self.P), _p(self.ks), _p(x), _p(out), C.c_int64(B)) == 0
        return out

    def mk_gate_nand(self, in0, in1, mode=MODE_FFT, nthreads=0):
        in0, in1 = _i32(in0), _i32(in1)
        B = in0.shape[0]
        out = np.zeros_like(in0)
        margin = C.c_double(0)
        rc = lib().orc_mk_gate_nand_batch(C.byref(self.P), C.c_int32(self.parties), _p(self.bk_re),
                                          _p(self.bk_im), _p(self.bk_i32), _p(self.ks), C.c_int32(mode),
                                          _p(in0), _p(in1), _p(out), C.c_int64(B), C.c_int32(nthreads),
                                          C.byref(margin))
        assert rc == 0
        self.last_margin = margin.value
        return out
