"""Multi-key TFHE keys on the host — mirrors src/mk_api.jl and the key-generation half of
src/mk_internals.jl (:101-139 SharedKey/PublicKey, :185-227 RGSW.UniEnc, :304-345 RGSW.Expand,
:419-461 bootstrap key).  Host-side, one-off, RNG-bound work (out of scope for the GPU);
the flat arrays produced here are what the C ABI's tfhe_mk_load_* entry points take.

Flat MK bootstrap key: int32 [P (party i)][n (bit j)][2*l*P + 2*l][N] with, per (i, j), the polys
x[l][P] | y[l][P] | c0[l] | c1[l] of MKTGswExpSample (mk_internals.jl:243-271) — x[p][q] at p*P + q.
"""
import numpy as np

from . import _lib
from .keys import SecretKey, TLweKey, make_keyswitch_key
from .lwe import lwe_phase
from .numeric import (decompose, dtot32, encode_message, negacyclic_mul_small, rand_gaussian_torus32,
                      rand_uniform_bool, rand_uniform_torus32, wrap32)
from .params import SchemeParameters


class SharedKey:
    """mk_api.jl:44-50 / mk_internals.jl:101-113 — l uniform polynomials shared by all parties."""

    def __init__(self, rng, params: SchemeParameters):
        self.params = params
        self.a = rand_uniform_torus32(rng, params.bs_decomp_length, params.tlwe_polynomial_degree)


def _gauss_poly(rng, alpha, *shape):
    return dtot32(rng.standard_normal(size=shape) * alpha).astype(np.int64)


class CloudKeyPart:
    """mk_api.jl:60-80 — what one party generates: public key b (mk_internals.jl:116-139), the
    uni-encryptions of its LWE key bits (mk_internals.jl:185-227, :419-439) and its keyswitch key."""

    def __init__(self, rng, secret_key: SecretKey, shared_key: SharedKey):
        p = secret_key.params
        self.params = p
        N, l, n = p.tlwe_polynomial_degree, p.bs_decomp_length, p.lwe_size
        alpha = p.bs_noise_stddev
        tlwe_key = TLweKey(rng, N, p.tlwe_mask_size)
        s = tlwe_key.key[0]                                             # mask_size = 1 (mk_internals.jl:129)
        a = shared_key.a
        # PublicKey: b_i = s * a_i + e_i                                 mk_internals.jl:131-136
        self.public_b = wrap32(negacyclic_mul_small(s, a).astype(np.int64) + _gauss_poly(rng, alpha, l, N))
        gadget = np.array([1 << (32 - (q + 1) * p.bs_log2_base) for q in range(l)], np.int64)
        m = secret_key.key.key.astype(np.int64)                         # the n messages (LWE key bits)
        # RGSW.UniEnc for all n bits at once                            mk_internals.jl:185-227
        r = rand_uniform_bool(rng, n, N)                                # :197
        c1 = rand_uniform_torus32(rng, n, l, N)                         # :200
        c0 = _gauss_poly(rng, alpha, n, l, N) + negacyclic_mul_small(s, c1).astype(np.int64)   # :202-206
        c0[:, :, 0] += m[:, None] * gadget[None, :]                     # + message * gadget (constant term)
        d1 = _gauss_poly(rng, alpha, n, l, N) + negacyclic_mul_small(r[:, None, :], a[None]).astype(np.int64)   # :209-213
        d1[:, :, 0] += m[:, None] * gadget[None, :]
        d0 = _gauss_poly(rng, alpha, n, l, N) + negacyclic_mul_small(r[:, None, :], self.public_b[None]).astype(np.int64)  # :214-217
        f1 = rand_uniform_torus32(rng, n, l, N)                         # :220
        f0 = _gauss_poly(rng, alpha, n, l, N) + negacyclic_mul_small(s, f1).astype(np.int64)   # :222-226
        f0 += r[:, None, :].astype(np.int64) * gadget[None, :, None]    # + r * gadget (every coefficient)
        self.c0, self.c1, self.d0, self.d1, self.f0, self.f1 = (wrap32(v) for v in (c0, c1, d0, d1, f0, f1))
        self.ks = make_keyswitch_key(rng, p.ks_noise_stddev, p.ks_decomp_length, p.ks_log2_base,
                                     secret_key.key, tlwe_key)          # mk_api.jl:74-76


class MKCloudKey:
    """mk_api.jl:83-101 — expands the parts (RGSW.Expand, mk_internals.jl:304-345) into the flat key.

    expand="host" (default): numpy, as the reference does it on the CPU; `.bootstrap_key` is the flat Int32 key.
    expand="device": the parts are kept as they are and every engine expands them on its GPU when it is created
    (tfhe_mk_expand_load_bootstrap_key) — what makes the full-size 4- and 8-party sets (mk_api.jl:16-34: up to
    3.6 M polynomial products, a 2.4 GB key) practical; `.bootstrap_key` is then fetched from the device on demand."""

    def __init__(self, ck_parts, expand="host"):
        p = ck_parts[0].params
        self.params = p
        self.parties = len(ck_parts)
        assert self.parties <= p.max_parties                            # mk_api.jl:94
        assert expand in ("host", "device")
        self.expand = expand
        self._parts = ck_parts
        self._bootstrap_key = self._expand_on_host() if expand == "host" else None
        self.keyswitch_key = np.stack([part.ks for part in ck_parts])   # [P][N][t][base-1][n+1]
        self._engines = {}

    def _expand_on_host(self):
        ck_parts = self._parts
        p, P = self.params, self.parties
        N, l, n, beta = p.tlwe_polynomial_degree, p.bs_decomp_length, p.lwe_size, p.bs_log2_base
        per = 2 * l * P + 2 * l
        bk = np.zeros((P, n, per, N), np.int32)
        pub = [part.public_b for part in ck_parts]
        for i, part in enumerate(ck_parts):                             # party i, all n bits at once
            x = np.zeros((n, l, P, N), np.int64)
            y = np.zeros((n, l, P, N), np.int64)
            for q in range(P):
                if q == i:
                    x[:, :, q] = part.d0                                # :328-329 (zero added)
                    y[:, :, q] = part.d1                                # :335-336
                    continue
                # g^{-1}(b_q[jj] - b_i[jj]) for jj = 1..l               :318-324
                dec = decompose(wrap32(pub[q].astype(np.int64) - pub[i].astype(np.int64)), l, beta)  # [u][jj][N]
                for jj in range(l):
                    xs = part.d0[:, jj].astype(np.int64)
                    ys = np.zeros((n, N), np.int64)
                    for u in range(l):
                        xs = xs + negacyclic_mul_small(dec[u, jj], part.f0[:, u])   # :331
                        ys = ys + negacyclic_mul_small(dec[u, jj], part.f1[:, u])   # :338
                    x[:, jj, q] = xs
                    y[:, jj, q] = ys
            bk[i, :, 0:l * P] = wrap32(x).reshape(n, l * P, N)
            bk[i, :, l * P:2 * l * P] = wrap32(y).reshape(n, l * P, N)
            bk[i, :, 2 * l * P:2 * l * P + l] = part.c0
            bk[i, :, 2 * l * P + l:] = part.c1
        return bk

    def _part_arrays(self):
        parts = self._parts
        return [np.stack([part.public_b for part in parts])] + [np.stack([getattr(part, name) for part in parts])
                                                               for name in ("c0", "c1", "d0", "d1", "f0", "f1")]

    @property
    def bootstrap_key(self):
        """Flat Int32 key [P][n][2lP + 2l][N] (host expansion, or downloaded from a device that expanded it)."""
        if self._bootstrap_key is None:
            e = _lib.Engine(self.params, 0)
            try:
                self._bootstrap_key = e.mk_expand_load_bootstrap_key(self.parties, *self._part_arrays(), want_expanded=True)
            finally:
                e.close()
        return self._bootstrap_key

    def engine(self, device=0):
        key = device if np.ndim(device) == 0 else tuple(int(d) for d in device)
        e = self._engines.get(key)
        if e is None:
            e = _lib.Engine(self.params, device) if np.ndim(device) == 0 else _lib.Engine(self.params, devices=list(key))
            if self.expand == "device":
                e.mk_expand_load_bootstrap_key(self.parties, *self._part_arrays())
            else:
                e.mk_load_bootstrap_key(self.bootstrap_key, self.parties)
            e.mk_load_keyswitch_key(self.keyswitch_key, self.parties)
            self._engines[key] = e
        return e

    def close(self):
        for e in self._engines.values():
            e.close()
        self._engines = {}


class MKLweSample:
    """mk_internals.jl:6-18.  Flat layout = a[:,0], a[:,1], ..., b (include/tfhe_mi355x.h)."""
    __slots__ = ("a", "b")

    def __init__(self, a, b):
        self.a = np.asarray(a, np.int32)        # [parties][n]
        self.b = np.int32(b)

    def flat(self):
        return np.concatenate([self.a.reshape(-1), np.array([self.b], np.int32)])


def mk_encrypt(rng, secret_keys, message):
    """mk_api.jl:111-126.  A bool -> flat int32 [P*n+1]; an array of bools -> int32 [B][P*n+1]."""
    p = secret_keys[0].params
    P, n = len(secret_keys), p.lwe_size
    bits = np.atleast_1d(np.asarray(message, bool))
    B = bits.size
    mu = np.where(bits, encode_message(1, 8), encode_message(-1, 8)).astype(np.int64)
    a = rand_uniform_torus32(rng, B, P, n)
    s = np.stack([sk.key.key for sk in secret_keys]).astype(np.int64)           # [P][n]
    noise = rand_gaussian_torus32(rng, 0, p.lwe_noise_stddev, B).astype(np.int64)
    b = wrap32(mu + noise + np.einsum("bpn,pn->b", a.astype(np.int64), s))
    flat = np.concatenate([a.reshape(B, P * n), b[:, None]], axis=1).astype(np.int32)
    return flat[0] if np.ndim(message) == 0 else flat


def mk_phase(secret_keys, flat):
    """mk_internals.jl:29-35 as written: b + sum_p lwe_phase((a_p, 0), s_p) = b - sum_p <a_p, s_p>."""
    flat = np.atleast_2d(np.asarray(flat, np.int32))
    P, n = len(secret_keys), secret_keys[0].params.lwe_size
    a = flat[:, :-1].reshape(-1, P, n).astype(np.int64)
    s = np.stack([sk.key.key for sk in secret_keys]).astype(np.int64)
    return wrap32(flat[:, -1].astype(np.int64) - np.einsum("bpn,pn->b", a, s))


def mk_decrypt(secret_keys, flat):
    """mk_api.jl:135-138"""
    ph = mk_phase(secret_keys, flat) > 0
    return bool(ph[0]) if np.ndim(flat) == 1 else ph


def mk_gate_nand(ck: MKCloudKey, x, y, device=0):
    """mk_gates.jl:7-12 — one batched call into the HIP engine; x, y flat samples or [B][P*n+1] matrices."""
    x2, y2 = np.atleast_2d(np.asarray(x, np.int32)), np.atleast_2d(np.asarray(y, np.int32))
    out = ck.engine(device).mk_gate_nand(x2, y2)
    return out[0] if np.ndim(x) == 1 else out
