"""ctypes binding of lib/libtfhe_mi355x.so (C ABI: include/tfhe_mi355x.h).

There is no CPU fallback: if the HIP library is missing or no device is usable, calls raise.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFHE_MI355X_LIB", os.path.join(_PKG, "lib", "libtfhe_mi355x.so"))   # override: A/B builds

# symbols include/tfhe_mi355x.h declares (tests check that the .so exports every one of them)
ABI_SYMBOLS = [
    "tfhe_abi_version", "tfhe_device_count", "tfhe_ctx_create", "tfhe_ctx_destroy", "tfhe_last_error",
    "tfhe_ctx_params", "tfhe_load_bootstrap_key_i32", "tfhe_load_bootstrap_key_c128",
    "tfhe_load_keyswitch_key", "tfhe_gates_batch", "tfhe_gates_batch_dev", "tfhe_bootstrap_batch",
    "tfhe_keyswitch_batch", "tfhe_mk_load_bootstrap_key_i32", "tfhe_mk_load_keyswitch_key",
    "tfhe_mk_gate_nand_batch", "tfhe_last_timing_ms", "tfhe_last_rotation_count", "tfhe_set_option",
    "tfhe_last_rounding_margin", "tfhe_wires_alloc", "tfhe_wires_upload", "tfhe_wires_download", "tfhe_gates_level",
    "tfhe_ctx_create_multi", "tfhe_ctx_device_count", "tfhe_shard_bounds", "tfhe_wires_gather",
    "tfhe_last_kernel_name", "tfhe_last_kernel_clock_mhz", "tfhe_mk_load_bootstrap_key_c128",
    "tfhe_mk_expand_load_bootstrap_key", "tfhe_keygen_cloud_key", "tfhe_host_alloc", "tfhe_host_free",
    "tfhe_timing_history_ms", "tfhe_gates_batch_submit", "tfhe_gates_batch_wait", "tfhe_last_device_count",
    "tfhe_get_option", "tfhe_ctx_synchronize",
]
ABI_VERSION = 7
ERR_NOMEM = 6

OPCODES = dict(NAND=0, OR=1, AND=2, XOR=3, XNOR=4, NOT=5, NOR=6, ANDNY=7, ANDYN=8, ORNY=9, ORYN=10,
               MUX=11, CONST0=12, CONST1=13, COPY=14)


class TfheParams(C.Structure):
    _fields_ = [(f, C.c_int32) for f in
                ("n", "N", "k", "bs_l", "bs_log2_base", "ks_t", "ks_log2_base", "parties")]


def pinned_empty(shape, dtype=np.int32):
    """A numpy array in page-locked host memory (tfhe_host_alloc): operands / results of the host-buffer batch calls held
    in such arrays cross PCIe as single DMA transfers instead of being staged through the runtime's bounce buffers.  The
    block is returned (tfhe_host_free) when the last view of the array is gone."""
    import weakref
    lib = load()
    dtype = np.dtype(dtype)
    count = int(np.prod(shape))
    nbytes = max(count * dtype.itemsize, 1)
    p = C.c_void_p()
    rc = lib.tfhe_host_alloc(nbytes, C.byref(p))
    if rc:
        raise MemoryError(f"tfhe_host_alloc({nbytes}) failed: {lib.tfhe_last_error(None).decode()}")
    buf = (C.c_char * nbytes).from_address(p.value)
    weakref.finalize(buf, lib.tfhe_host_free, p.value)          # numpy holds `buf` for as long as any view lives
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


class EngineError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"tfhe_mi355x error {code}: {message}")
        self.code = code


_lib = None
_warned_inexact = set()


def _warn_inexact_once(params, margin, bound_log2):
    key = params.engine_tuple()
    if key in _warned_inexact:
        return
    _warned_inexact.add(key)
    import warnings
    warnings.warn(f"TFHE parameter set {key} (lwe_size, N, k, l, log2 Bg, t, log2 ks_base, parties) is outside the Float64 exactness "
                  f"domain: predicted rounding margin {margin:.2f} (a flipped rounding needs 0.5), worst-case magnitude 2^{bound_log2:.1f}. "
                  "The engine computes as the reference does, but result words may differ from the exact negacyclic product; "
                  "tfhe_last_rounding_margin measures the actual margin.", RuntimeWarning, stacklevel=3)


def load():
    """Loads the HIP library; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C tfhe.jl_amd/csrc` (hipcc, gfx950). The engine has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.tfhe_abi_version.restype = i32
    lib.tfhe_device_count.restype = i32
    lib.tfhe_ctx_create.argtypes = [C.POINTER(TfheParams), i32, C.POINTER(vp)]
    lib.tfhe_ctx_destroy.argtypes = [vp]
    lib.tfhe_ctx_destroy.restype = None
    lib.tfhe_last_error.argtypes = [vp]
    lib.tfhe_last_error.restype = C.c_char_p
    lib.tfhe_ctx_params.argtypes = [vp, C.POINTER(TfheParams)]
    lib.tfhe_load_bootstrap_key_i32.argtypes = [vp, vp]
    lib.tfhe_load_bootstrap_key_c128.argtypes = [vp, vp]
    lib.tfhe_load_keyswitch_key.argtypes = [vp, vp]
    lib.tfhe_keygen_cloud_key.argtypes = [vp, vp, vp, C.c_double, C.c_double, vp, vp, vp]
    if hasattr(lib, "tfhe_host_alloc"):
        lib.tfhe_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
        lib.tfhe_host_free.argtypes = [vp]
        lib.tfhe_host_free.restype = None
    lib.tfhe_gates_batch.argtypes = [vp, vp, vp, vp, vp, vp, i64]
    lib.tfhe_gates_batch_dev.argtypes = [vp, vp, vp, vp, vp, vp, i64, vp]
    if hasattr(lib, "tfhe_gates_batch_submit"):
        lib.tfhe_gates_batch_submit.argtypes = [vp, vp, vp, vp, vp, vp, i64, C.POINTER(i32)]
        lib.tfhe_gates_batch_wait.argtypes = [vp, i32]
    lib.tfhe_bootstrap_batch.argtypes = [vp, i32, vp, vp, i64, i32]
    lib.tfhe_keyswitch_batch.argtypes = [vp, vp, vp, i64]
    lib.tfhe_mk_load_bootstrap_key_i32.argtypes = [vp, vp, i32]
    lib.tfhe_mk_load_keyswitch_key.argtypes = [vp, vp, i32]
    lib.tfhe_mk_load_bootstrap_key_c128.argtypes = [vp, vp, i32]
    lib.tfhe_mk_expand_load_bootstrap_key.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tfhe_mk_gate_nand_batch.argtypes = [vp, vp, vp, vp, i64]
    lib.tfhe_last_timing_ms.argtypes = [vp, i32, C.POINTER(C.c_float)]
    if hasattr(lib, "tfhe_timing_history_ms"):
        lib.tfhe_timing_history_ms.argtypes = [vp, i32, C.POINTER(C.c_float), i32, C.POINTER(i32)]
    lib.tfhe_last_rotation_count.argtypes = [vp]
    lib.tfhe_last_rotation_count.restype = i64
    lib.tfhe_last_device_count.argtypes = [vp]
    lib.tfhe_last_device_count.restype = i32
    lib.tfhe_set_option.argtypes = [vp, C.c_char_p, i64]
    lib.tfhe_get_option.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    lib.tfhe_last_rounding_margin.argtypes = [vp, C.POINTER(C.c_double)]
    lib.tfhe_wires_alloc.argtypes = [vp, i64]
    lib.tfhe_wires_upload.argtypes = [vp, i64, i64, vp]
    lib.tfhe_wires_download.argtypes = [vp, i64, i64, vp]
    lib.tfhe_gates_level.argtypes = [vp, vp, vp, vp, vp, vp, i64]
    lib.tfhe_ctx_create_multi.argtypes = [C.POINTER(TfheParams), vp, i32, C.POINTER(vp)]
    lib.tfhe_ctx_device_count.argtypes = [vp]
    lib.tfhe_ctx_device_count.restype = i32
    lib.tfhe_shard_bounds.argtypes = [vp, i64, i32, vp]
    lib.tfhe_wires_gather.argtypes = [vp, vp, i64, vp]
    lib.tfhe_last_kernel_name.argtypes = [vp]
    lib.tfhe_last_kernel_name.restype = C.c_char_p
    lib.tfhe_last_kernel_clock_mhz.argtypes = [vp, C.POINTER(C.c_double)]
    if hasattr(lib, "tfhe_ctx_synchronize"):
        lib.tfhe_ctx_synchronize.argtypes = [vp]
    ver = lib.tfhe_abi_version()
    if ver < 0 and not os.environ.get("TFHE_MI355X_ALLOW_EXPERIMENT"):
        # a development build (-DTFHE_EXPERIMENT: in-kernel stamps, environment-variable overrides; csrc/experiment.hpp)
        raise ImportError(f"{LIB_PATH} is a development build (ABI version {ver}): set TFHE_MI355X_ALLOW_EXPERIMENT=1 to load it knowingly")
    if abs(ver) != ABI_VERSION and not os.environ.get("TFHE_MI355X_ALLOW_ABI_MISMATCH"):     # (the override is for A/B runs against an older build)
        raise ImportError(f"{LIB_PATH} has ABI version {ver}, this package needs {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i32c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


def shard_bounds(opcodes, shards):
    """The library's own sharding rule (tfhe_shard_bounds): list of (start, end) per shard."""
    ops = None if opcodes is None else np.ascontiguousarray(opcodes, dtype=np.uint8)
    B = 0 if ops is None else ops.size
    out = np.zeros(int(shards) + 1, np.int64)
    rc = load().tfhe_shard_bounds(_ptr(ops), B, int(shards), _ptr(out))
    if rc != 0:
        raise EngineError(rc, "tfhe_shard_bounds: invalid argument")
    return [(int(out[r]), int(out[r + 1])) for r in range(int(shards))]


class Engine:
    """One context (tfhe_ctx): keys resident on one GPU — or, with `devices=[...]`, replicated on several, every
    host-buffer batch call fanned out over them inside the library (tfhe_ctx_create_multi)."""

    def __init__(self, params, device=0, devices=None):
        self._lib = load()
        self.params = params
        n, N, k, l, b, t, g, parties = params.engine_tuple()
        self.n, self.N, self.k, self.parties = n, N, k, parties
        p = TfheParams(n, N, k, l, b, t, g, parties)
        h = C.c_void_p()
        if devices is None:
            rc = self._lib.tfhe_ctx_create(C.byref(p), int(device), C.byref(h))
            self.devices = [int(device)]
        else:
            ids = np.ascontiguousarray(devices, dtype=np.int32)
            rc = self._lib.tfhe_ctx_create_multi(C.byref(p), _ptr(ids), ids.size, C.byref(h))
            self.devices = [int(v) for v in ids]
        if rc != 0:
            raise EngineError(rc, self._lib.tfhe_last_error(None).decode())
        self._h = h
        self._in_flight = {}
        self.device = self.devices[0]
        # is this parameter set inside what a Float64 transform computes exactly?  (decided by tfhe_ctx_create from the parameters
        # alone; the reference warns about the same thing, polynomials.jl:135-144)
        self.exact_domain = self.get_option("exact_domain")
        if self.exact_domain == 0:
            _warn_inexact_once(params, self.get_option("exact_margin_x1e6") / 1e6, self.get_option("exact_bound_log2_x1000") / 1e3)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tfhe_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(rc, self._lib.tfhe_last_error(self._h).decode())

    # ---- keys ----
    def load_bootstrap_key(self, bk_i32):
        bk = _i32c(bk_i32)
        l = self.params.bs_decomp_length
        want = self.n * l * (self.k + 1) ** 2 * self.N
        if bk.size != want:
            raise ValueError(f"bootstrap key has {bk.size} words, expected {want}")
        self._check(self._lib.tfhe_load_bootstrap_key_i32(self._h, _ptr(bk)))

    def load_bootstrap_key_spectra(self, spectra):
        s = np.ascontiguousarray(spectra, dtype=np.complex128)
        l = self.params.bs_decomp_length
        want = self.n * l * (self.k + 1) ** 2 * (self.N // 2)
        if s.size != want:
            raise ValueError(f"bootstrap key spectra have {s.size} values, expected {want}")
        self._check(self._lib.tfhe_load_bootstrap_key_c128(self._h, _ptr(s)))

    def load_keyswitch_key(self, ks):
        ks = _i32c(ks)
        p = self.params
        want = self.k * self.N * p.ks_decomp_length * ((1 << p.ks_log2_base) - 1) * (self.n + 1)
        if ks.size != want:
            raise ValueError(f"keyswitch key has {ks.size} words, expected {want}")
        self._check(self._lib.tfhe_load_keyswitch_key(self._h, _ptr(ks)))

    def keygen_cloud_key(self, lwe_key, tlwe_key, bs_noise_stddev, ks_noise_stddev, seed, want_arrays=True):
        """Generates bootstrap + keyswitch key on the device and loads them (tfhe_keygen_cloud_key).  `seed`: six 32-bit
        words; words 0-1 key the (public) mask streams, words 2-5 are the 128-bit noise key and AS SECRET AS THE SECRET KEY.
        Returns the canonical Int32 arrays (bk [n][l][k+1][k+1][N], ks [kN][t][base-1][n+1]) unless want_arrays is False."""
        lwe_key, tlwe_key = _i32c(lwe_key), _i32c(tlwe_key)
        seed = np.ascontiguousarray(seed, dtype=np.uint32).reshape(-1)
        if seed.size != 6:
            raise ValueError("keygen_cloud_key: seed must be six 32-bit words (2 for the masks, 4 secret ones for the noise)")
        p = self.params
        if lwe_key.size != self.n or tlwe_key.size != self.k * self.N:
            raise ValueError(f"secret keys have {lwe_key.size} / {tlwe_key.size} words, expected {self.n} / {self.k * self.N}")
        bk = ks = None
        if want_arrays:
            bk = np.empty((self.n, p.bs_decomp_length, self.k + 1, self.k + 1, self.N), np.int32)
            ks = np.empty((self.k * self.N, p.ks_decomp_length, (1 << p.ks_log2_base) - 1, self.n + 1), np.int32)
        self._check(self._lib.tfhe_keygen_cloud_key(self._h, _ptr(lwe_key), _ptr(tlwe_key), float(bs_noise_stddev),
                                                    float(ks_noise_stddev), _ptr(seed), _ptr(bk), _ptr(ks)))
        return bk, ks

    # ---- hot path, host buffers ----
    def gates(self, opcodes, in0, in1=None, in2=None, out=None):
        """`out`: an int32 [B][n+1] array to receive the result (e.g. one from pinned_empty, reused across calls);
        a fresh array otherwise."""
        ops = np.ascontiguousarray(opcodes, dtype=np.uint8)
        B = ops.size
        in0, in1, in2 = _i32c(in0), _i32c(in1), _i32c(in2)
        for a in (in0, in1, in2):
            if a is not None and a.shape != (B, self.n + 1):
                raise ValueError(f"operand shape {a.shape}, expected {(B, self.n + 1)}")
        if out is None:
            out = np.empty((B, self.n + 1), np.int32)
        elif out.dtype != np.int32 or out.shape != (B, self.n + 1) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous int32 array of shape {(B, self.n + 1)}")
        self._check(self._lib.tfhe_gates_batch(self._h, _ptr(ops), _ptr(in0), _ptr(in1), _ptr(in2), _ptr(out), B))
        return out

    def gates_submit(self, opcodes, in0, in1=None, in2=None, out=None):
        """Streaming form of gates(): enqueues the batch and returns (ticket, out) at once; `out` is complete after
        gates_wait(ticket).  Two batches may be in flight (the upload of one under the kernels of the other); operands
        and `out` should come from pinned_empty and must not be touched until the wait.  Arrays passed here are kept
        alive until then; an operand that is not already a C-contiguous int32 array is copied first."""
        ops = np.ascontiguousarray(opcodes, dtype=np.uint8)
        B = ops.size
        in0, in1, in2 = _i32c(in0), _i32c(in1), _i32c(in2)
        for a in (in0, in1, in2):
            if a is not None and a.shape != (B, self.n + 1):
                raise ValueError(f"operand shape {a.shape}, expected {(B, self.n + 1)}")
        if out is None:
            out = pinned_empty((B, self.n + 1))
        elif out.dtype != np.int32 or out.shape != (B, self.n + 1) or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous int32 array of shape {(B, self.n + 1)}")
        ticket = C.c_int32(-1)
        self._check(self._lib.tfhe_gates_batch_submit(self._h, _ptr(ops), _ptr(in0), _ptr(in1), _ptr(in2), _ptr(out), B, C.byref(ticket)))
        if ticket.value in (0, 1):
            self._in_flight[ticket.value] = (in0, in1, in2, out)      # a displaced batch has been waited for by the library
        return ticket.value, out

    def gates_wait(self, ticket):
        self._check(self._lib.tfhe_gates_batch_wait(self._h, int(ticket)))
        self._in_flight.pop(int(ticket), None)

    def synchronize(self):
        """Blocks until everything queued on the context has completed (tfhe_ctx_synchronize): callable from any thread, also
        while another thread is inside a call on this engine."""
        rc = self._lib.tfhe_ctx_synchronize(self._h)
        if rc != 0:
            raise EngineError(rc, "tfhe_ctx_synchronize failed")

    def gates_dev(self, opcodes, d_in0, d_in1, d_in2, d_out, B, stream=0):
        """Device-pointer variant: operands are integer device addresses (e.g. torch tensor .data_ptr())."""
        ops = np.ascontiguousarray(opcodes, dtype=np.uint8)
        assert ops.size == B
        self._check(self._lib.tfhe_gates_batch_dev(self._h, _ptr(ops), d_in0 or None, d_in1 or None,
                                                   d_in2 or None, d_out, B, stream or None))

    def bootstrap(self, mu, x, with_keyswitch=True):
        x = _i32c(x)
        if x.ndim != 2 or x.shape[1] != self.n + 1:
            raise ValueError(f"bootstrap input must be [B][{self.n + 1}], got {x.shape}")
        B = x.shape[0]
        width = self.n + 1 if with_keyswitch else self.k * self.N + 1
        out = np.empty((B, width), np.int32)
        self._check(self._lib.tfhe_bootstrap_batch(self._h, int(mu), _ptr(x), _ptr(out), B, 1 if with_keyswitch else 0))
        return out

    def keyswitch(self, x):
        x = _i32c(x)
        B = x.shape[0]
        if x.ndim != 2 or x.shape[1] != self.k * self.N + 1:
            raise ValueError("keyswitch input must be [B][k*N+1]")
        out = np.empty((B, self.n + 1), np.int32)
        self._check(self._lib.tfhe_keyswitch_batch(self._h, _ptr(x), _ptr(out), B))
        return out

    # ---- levelised circuits on the device-resident wire table ----
    def wires_alloc(self, num_wires):
        self._check(self._lib.tfhe_wires_alloc(self._h, int(num_wires)))

    def wires_upload(self, first, samples):
        m = _i32c(samples)
        if m.ndim != 2 or m.shape[1] != self.n + 1:
            raise ValueError("samples must be [count][n+1]")
        self._check(self._lib.tfhe_wires_upload(self._h, int(first), m.shape[0], _ptr(m)))

    def wires_download(self, first, count):
        out = np.empty((int(count), self.n + 1), np.int32)
        self._check(self._lib.tfhe_wires_download(self._h, int(first), int(count), _ptr(out)))
        return out

    def wires_gather(self, wires):
        """Rows of the given wire indices, in one device gather + one copy."""
        idx = np.ascontiguousarray(wires, dtype=np.int32).reshape(-1)
        out = np.empty((idx.size, self.n + 1), np.int32)
        self._check(self._lib.tfhe_wires_gather(self._h, _ptr(idx), idx.size, _ptr(out)))
        return out

    def gates_level(self, opcodes, a, b, c, out):
        ops = np.ascontiguousarray(opcodes, dtype=np.uint8)
        arrs = [None if v is None else _i32c(v) for v in (a, b, c, out)]
        for v in arrs:
            if v is not None and v.shape != (ops.size,):
                raise ValueError("index arrays must have one entry per gate")
        self._check(self._lib.tfhe_gates_level(self._h, _ptr(ops), _ptr(arrs[0]), _ptr(arrs[1]), _ptr(arrs[2]),
                                               _ptr(arrs[3]), ops.size))

    # ---- multi-key ----
    def mk_load_bootstrap_key(self, bk_i32, parties):
        bk = _i32c(bk_i32)
        P, l = int(parties), self.params.bs_decomp_length
        want = P * self.n * (2 * l * P + 2 * l) * self.N
        if bk.size != want:
            raise ValueError(f"multi-key bootstrap key has {bk.size} words, expected {want} for {P} parties")
        self._check(self._lib.tfhe_mk_load_bootstrap_key_i32(self._h, _ptr(bk), P))
        self._mk_parties = P

    def mk_load_bootstrap_key_spectra(self, spectra, parties):
        """The reference's stored form: complex128 [P][n][2lP + 2l][N/2]."""
        sp = np.ascontiguousarray(spectra, dtype=np.complex128)
        P, l = int(parties), self.params.bs_decomp_length
        want = P * self.n * (2 * l * P + 2 * l) * (self.N // 2)
        if sp.size != want:
            raise ValueError(f"multi-key bootstrap spectra have {sp.size} values, expected {want} for {P} parties")
        self._check(self._lib.tfhe_mk_load_bootstrap_key_c128(self._h, _ptr(sp), P))
        self._mk_parties = P

    def mk_expand_load_bootstrap_key(self, parties, pub_b, c0, c1, d0, d1, f0, f1, want_expanded=False):
        """RGSW.Expand on the device (tfhe_mk_expand_load_bootstrap_key).  pub_b: [P][l][N]; c0 .. f1: [P][n][l][N].
        Returns the expanded Int32 key [P][n][2lP + 2l][N] if want_expanded."""
        P, l = int(parties), self.params.bs_decomp_length
        arrs = [_i32c(a) for a in (pub_b, c0, c1, d0, d1, f0, f1)]
        if arrs[0].shape != (P, l, self.N):
            raise ValueError(f"public keys must be [{P}][{l}][{self.N}], got {arrs[0].shape}")
        for a in arrs[1:]:
            if a.shape != (P, self.n, l, self.N):
                raise ValueError(f"uni-encryption arrays must be [{P}][{self.n}][{l}][{self.N}], got {a.shape}")
        out = np.empty((P, self.n, 2 * l * P + 2 * l, self.N), np.int32) if want_expanded else None
        self._check(self._lib.tfhe_mk_expand_load_bootstrap_key(self._h, P, *[_ptr(a) for a in arrs], _ptr(out)))
        self._mk_parties = P
        return out

    def mk_load_keyswitch_key(self, ks, parties):
        ks = _i32c(ks)
        P, p = int(parties), self.params
        want = P * self.N * p.ks_decomp_length * ((1 << p.ks_log2_base) - 1) * (self.n + 1)
        if ks.size != want:
            raise ValueError(f"multi-key keyswitch key has {ks.size} words, expected {want} for {P} parties")
        self._check(self._lib.tfhe_mk_load_keyswitch_key(self._h, _ptr(ks), P))

    def mk_gate_nand(self, in0, in1):
        in0, in1 = _i32c(in0), _i32c(in1)
        P = getattr(self, "_mk_parties", None)
        if P is None:
            raise EngineError(3, "mk_gate_nand: multi-key keys not loaded")
        if in0.ndim != 2 or in0.shape[1] != P * self.n + 1 or in1.shape != in0.shape:
            raise ValueError(f"multi-key operands must both be [B][{P * self.n + 1}], got {in0.shape} and {in1.shape}")
        out = np.empty_like(in0)
        self._check(self._lib.tfhe_mk_gate_nand_batch(self._h, _ptr(in0), _ptr(in1), _ptr(out), in0.shape[0]))
        return out

    # ---- measurement ----
    def last_timing_ms(self, which):
        ms = C.c_float(0)
        self._check(self._lib.tfhe_last_timing_ms(self._h, int(which), C.byref(ms)))
        return float(ms.value)

    def timing_history_ms(self, which, max_calls=32):
        """Kernel-side durations (HIP events) of up to the last 32 batch calls, oldest first, read after the calls in one go
        (tfhe_timing_history_ms): no synchronisation between the calls themselves."""
        buf = (C.c_float * int(max_calls))()
        n = C.c_int32(0)
        self._check(self._lib.tfhe_timing_history_ms(self._h, int(which), buf, int(max_calls), C.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def last_rotation_count(self):
        return int(self._lib.tfhe_last_rotation_count(self._h))

    def last_device_count(self):
        """Device contexts the last batch / level call was sharded over (1 on a one-device context)."""
        return int(self._lib.tfhe_last_device_count(self._h))

    def last_rounding_margin(self):
        m = C.c_double(0)
        self._check(self._lib.tfhe_last_rounding_margin(self._h, C.byref(m)))
        return float(m.value)

    def last_kernel_clock_mhz(self):
        m = C.c_double(0)
        self._check(self._lib.tfhe_last_kernel_clock_mhz(self._h, C.byref(m)))
        return float(m.value)

    def last_kernel_name(self):
        return self._lib.tfhe_last_kernel_name(self._h).decode()

    def device_count(self):
        return int(self._lib.tfhe_ctx_device_count(self._h))

    def set_option(self, name, value):
        self._check(self._lib.tfhe_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int64(0)
        self._check(self._lib.tfhe_get_option(self._h, name.encode(), C.byref(v)))
        return v.value
