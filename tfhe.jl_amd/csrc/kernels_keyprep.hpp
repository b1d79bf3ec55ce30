// kernels_keyprep.hpp — bootstrapping-key preparation at load time (Int32 / c128 -> engine-order spectra) and RGSW.Expand on the device
// (mk_internals.jl:304-345); emitted by engine_keys.hip only.
#pragma once
#include "kernels_common.hpp"
#include "kernels_n2048.hpp"

#ifdef TFHE_EMIT_KEYPREP_KERNELS       // (defined by engine_keys.hip, the one translation unit that launches them)
// key preparation for N = 2048: Int32 polynomial -> [wave][8][64] spectra scaled by 1/1024
__global__ __launch_bounds__(128) void bk_prepare_kernel_n2048(const int32_t *__restrict__ bk_i32, cplx *__restrict__ out,
                                                             const cplx *__restrict__ tw1f2, const cplx *__restrict__ tw2)
{
    __shared__ __attribute__((aligned(16))) cplx xch_all[2 * kXchElems + 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const bool wave1 = (tid >> 6) != 0;
    cplx *xch = xch_all + (wave1 ? kXchElems : 0);
    cplx *tw2_lds = xch_all + 2 * kXchElems;
    const size_t q = blockIdx.x;
    const int32_t *poly = bk_i32 + q * kN2;
    const double sg = wave1 ? -0.70710678118654752440 : 0.70710678118654752440;
    cplx tw1f[8];
#pragma unroll
    for (int qq = 0; qq < 8; qq++) tw1f[qq] = tw1f2[(wave1 ? 512 : 0) + qq * 64 + lane];
    if (tid < 64) tw2_lds[tid] = tw2[tid];
    __syncthreads();
    cplx x[8];
    static_for<0, 8>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        const double lo = (double)poly[lane + 64 * R], l2 = (double)poly[lane + 64 * R + 512];
        const double hi = (double)poly[lane + 64 * R + 1024], h2 = (double)poly[lane + 64 * R + 1536];
        x[R] = fwd_in_2048<R>(lo, hi, l2 - h2, l2 + h2, sg, wave1);
    });
    fft_fwd_half(lane, x, tw1f, tw2_lds, xch);
    const double s = 1.0 / 1024.0;
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) out[q * 2 * kM + (wave1 ? kM : 0) + k2 * 64 + lane] = mk(x[k2].x * s, x[k2].y * s);
}

// the reference's spectra for N = 2048 (natural order, 1024 values) -> engine order
__global__ __launch_bounds__(128) void bk_permute_c128_kernel_n2048(const cplx *__restrict__ in, cplx *__restrict__ out)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t q = blockIdx.x;
    const double s = 1.0 / 1024.0;
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
        const cplx v = in[q * 2 * kM + 2 * freq_of(lane, k2) + wv];
        out[q * 2 * kM + wv * kM + k2 * 64 + lane] = mk(v.x * s, v.y * s);
    }
}

// Bootstrapping-key preparation: Int32 polynomial -> spectrum in the engine's order, scaled by `scale`
// (1/M for key polynomials: the analogue of forward_transform.(bk), bootstrap.jl:12; 1 for a multiplier polynomial).
__global__ __launch_bounds__(64) void bk_prepare_kernel(const int32_t *__restrict__ bk_i32, cplx *__restrict__ out, Tables T, double scale = 1.0 / kM)
{
    __shared__ __attribute__((aligned(16))) cplx xch[kXchElems];
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    cplx x[8];
    load_poly(lane, bk_i32 + q * kN, T, x);
    fwd_pass_a(lane, x, T);
    x1_store_a(lane, x, xch);
    __syncthreads();
    x1_load_b(lane, x, xch);
    __syncthreads();
    fwd_pass_b(lane, x, T);
    x2_store(lane, x, xch);
    __syncthreads();
    x2_load(lane, x, xch);
    __syncthreads();
    fwd_pass_c(x);
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) out[q * kM + k2 * 64 + lane] = mk(x[k2].x * scale, x[k2].y * scale);
}

// ---- RGSW.Expand on the device (mk_internals.jl:304-345) ---------------------------------------------------------
// For party i, bit j, row jj and every other party q:
//     x[jj, q] = d0[jj] + sum_u  g^-1(b_q[jj] - b_i[jj])[u] (*) f0[u]          y[jj, q] = sum_u g^-1(...)[u] (*) f1[u]
// ((*) = negacyclic product mod 2^32, one factor a decomposition digit polynomial: exact through the Float64
// transform exactly as in the external product).  The digit polynomials depend on (q, jj, u) only and f0 / f1 on
// (j, u) only, so both are transformed once and an output polynomial is l spectrum products, one inverse transform,
// one rounding.  One wave per output polynomial; the result is written in the flat Int32 key layout
// [n][2 l P + 2 l][N] of party i (include/tfhe_mi355x.h), which bk_prepare_kernel then turns into engine spectra.
struct MkExpandArgs {
    const cplx *dec;      // [P-1 (other party, in order)][l (u)][l (jj)][8][64]   digit spectra, unscaled
    const cplx *f;        // [2 (f0 | f1)][n][l (u)][8][64]                          spectra scaled 1/M
    const int32_t *d0;    // [n][l][N]
    int32_t *key;         // [n][2 l P + 2 l][N]   (x and y slots of the OTHER parties are written here)
    Tables T;
    int32_t n, l, parties, party;
};

__global__ __launch_bounds__(64) void mk_expand_kernel(MkExpandArgs A)
{
    __shared__ __attribute__((aligned(16))) cplx xch[kXchElems + 64];
    cplx *tw2_lds = xch + kXchElems;
    const int lane = threadIdx.x;
    // grid: x = bit j, y = (jj, other-party index oq), z = 0 (x) | 1 (y)
    const int j = blockIdx.x, jj = blockIdx.y % A.l, oq = blockIdx.y / A.l, xy = blockIdx.z;
    const int q = oq < A.party ? oq : oq + 1;                        // the oq-th party other than `party`
    const int per = 2 * A.l * A.parties + 2 * A.l;
    cplx tw1f[8];
#pragma unroll
    for (int k = 0; k < 8; k++) tw1f[k] = A.T.tw1f[k * 64 + lane];
    tw2_lds[lane] = A.T.tw2[lane];
    cplx acc[8];
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) acc[k2] = mk(0.0, 0.0);
    for (int u = 0; u < A.l; u++) {
        const cplx *dp = A.dec + ((size_t)(oq * A.l + u) * A.l + jj) * kM + lane;
        const cplx *fp = A.f + (((size_t)xy * A.n + j) * A.l + u) * kM + lane;
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) acc[k2] = cfma(dp[k2 * 64], fp[k2 * 64], acc[k2]);
    }
    WAVE_LDS_FENCE();
    fft_inv_wave(lane, acc, tw1f, tw2_lds, xch);
    int32_t r[16];
#pragma unroll
    for (int m = 0; m < 16; m++) r[m] = xy == 0 ? A.d0[((size_t)j * A.l + jj) * kN + lane + 64 * m] : 0;     // y has no d1 term for q != party (:334-339)
    untwist_add2(acc, r);
    int32_t *o = A.key + ((size_t)j * per + (xy == 0 ? 0 : A.l * A.parties) + jj * A.parties + q) * kN;
#pragma unroll
    for (int m = 0; m < 16; m++) o[lane + 64 * m] = r[m];
}

// the party's own columns and the c0 / c1 rows are copies: x[jj, party] = d0[jj], y[jj, party] = d1[jj]  (:328-336)
__global__ void mk_expand_copy_kernel(const int32_t *__restrict__ c0, const int32_t *__restrict__ c1, const int32_t *__restrict__ d0,
                                      const int32_t *__restrict__ d1, int32_t *__restrict__ key, int n, int l, int parties, int party, int N = kN)
{
    const int j = blockIdx.x, jj = blockIdx.y, which = blockIdx.z;       // which: 0 x, 1 y, 2 c0, 3 c1
    const int per = 2 * l * parties + 2 * l;
    const int32_t *src = (which == 0 ? d0 : which == 1 ? d1 : which == 2 ? c0 : c1) + ((size_t)j * l + jj) * N;
    const int slot = which == 0 ? jj * parties + party : which == 1 ? l * parties + jj * parties + party : which == 2 ? 2 * l * parties + jj : 2 * l * parties + l + jj;
    int32_t *dst = key + ((size_t)j * per + slot) * N;
    for (int t = threadIdx.x; t < N; t += blockDim.x) dst[t] = src[t];
}

// The reference's stored spectra (natural frequency order, polynomials.jl:106-112) -> engine order.
__global__ __launch_bounds__(64) void bk_permute_c128_kernel(const cplx *__restrict__ in, cplx *__restrict__ out)
{
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    const double s = 1.0 / kM;
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) {
        const cplx v = in[q * kM + freq_of(lane, k2)];
        out[q * kM + k2 * 64 + lane] = mk(v.x * s, v.y * s);
    }
}
#endif  // TFHE_EMIT_KEYPREP_KERNELS
