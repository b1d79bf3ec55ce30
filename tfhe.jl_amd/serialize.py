"""Flat on-disk format for cloud keys and LWE batches.

The reference has no (de)serialisation at all (keys live only as Julia object graphs); this is the
engine's own versioned container so that keys minted elsewhere — e.g. by real TFHE.jl through
julia/TFHEMI355X/src/TFHEMI355X.jl's flattening — can be replayed on a GPU box.  Layout (little endian):

    magic  b"TFHEMI355X\\0"  (11 bytes) | version u32 | n_sections u32
    per section:  name (16 bytes, NUL padded) | dtype code u32 (0 = int32, 1 = float64, 2 = complex128, 3 = uint8)
                  | ndim u32 | shape u64[ndim] | raw C-order data

Sections of a cloud key: "params" (int32[8] as tfhe_params + float64 noise fields in "noise"),
"bootstrap_key" (int32 [n][l][k+1][k+1][N], or "bk_spectra" complex128 [n][l][k+1][k+1][N/2]),
"keyswitch_key" (int32 [kN][t][base-1][n+1]).
"""
import struct

import numpy as np

from .params import SchemeParameters

MAGIC = b"TFHEMI355X\0"
VERSION = 1
_DTYPES = {0: np.int32, 1: np.float64, 2: np.complex128, 3: np.uint8}
_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


def write_sections(path, sections):
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<II", VERSION, len(sections)))
        for name, arr in sections.items():
            a = np.ascontiguousarray(arr)
            if a.dtype not in _CODES:
                raise TypeError(f"section {name}: unsupported dtype {a.dtype}")
            nb = name.encode()
            if len(nb) > 16:
                raise ValueError("section name longer than 16 bytes")
            f.write(nb.ljust(16, b"\0"))
            f.write(struct.pack("<II", _CODES[a.dtype], a.ndim))
            f.write(struct.pack(f"<{a.ndim}Q", *a.shape))
            f.write(a.tobytes())


def read_sections(path):
    out = {}
    with open(path, "rb") as f:
        if f.read(len(MAGIC)) != MAGIC:
            raise ValueError("not a TFHEMI355X file")
        version, count = struct.unpack("<II", f.read(8))
        if version != VERSION:
            raise ValueError(f"unsupported version {version}")
        for _ in range(count):
            name = f.read(16).rstrip(b"\0").decode()
            code, ndim = struct.unpack("<II", f.read(8))
            shape = struct.unpack(f"<{ndim}Q", f.read(8 * ndim))
            dt = np.dtype(_DTYPES[code])
            nbytes = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
            buf = f.read(nbytes)
            if len(buf) != nbytes:
                raise ValueError(f"section {name} truncated")
            out[name] = np.frombuffer(buf, dt).reshape(shape).copy()
    return out


def save_cloud_key(path, ck):
    p = ck.params
    sections = {
        "params": np.array(p.engine_tuple(), np.int32),
        "noise": np.array([p.lwe_noise_stddev, p.bs_noise_stddev, p.ks_noise_stddev], np.float64),
        "bootstrap_key": ck.bootstrap_key,
        "keyswitch_key": ck.keyswitch_key,
    }
    write_sections(path, sections)


class LoadedCloudKey:
    """A cloud key read back from disk: same `.params`, `.bootstrap_key`, `.keyswitch_key`, `.engine()` surface
    as keys.CloudKey (so the gate_* functions take it unchanged)."""

    def __init__(self, sections):
        n, N, k, l, b, t, g, parties = [int(v) for v in sections["params"]]
        noise = sections.get("noise", np.zeros(3))
        self.params = SchemeParameters(n, float(noise[0]), N, k, l, b, float(noise[1]), t, g, float(noise[2]), parties)
        self.bootstrap_key = sections.get("bootstrap_key")
        self.bootstrap_spectra = sections.get("bk_spectra")
        self.keyswitch_key = sections["keyswitch_key"]
        self._engines = {}

    def engine(self, device=0):
        from . import _lib
        key = device if np.ndim(device) == 0 else tuple(int(d) for d in device)
        e = self._engines.get(key)
        if e is None:
            e = _lib.Engine(self.params, device) if np.ndim(device) == 0 else _lib.Engine(self.params, devices=list(key))
            if self.bootstrap_key is not None:
                e.load_bootstrap_key(self.bootstrap_key)
            else:
                e.load_bootstrap_key_spectra(self.bootstrap_spectra)
            e.load_keyswitch_key(self.keyswitch_key)
            self._engines[key] = e
        return e

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}


def load_cloud_key(path):
    return LoadedCloudKey(read_sections(path))
