// kernels_anyn.hpp — blind rotation for EVERY parameter set the reference would accept: any power-of-two polynomial degree
// N >= 2, any mask size k, any decomposition (l, beta) with l * beta <= 32; in the multi-key form any number of parties.
//
// SchemeParameters is an unvalidated positional struct (api.jl:4-21) and the reference's transform works for any even length
// (polynomials.jl:44-58,106-132): a parameter set it accepts must not be refused here because no tuned kernel was written
// for it.  The tuned kernels (kernels_blind_rotate.hpp and the family headers it lists) are built around N = 1024 / 2048 as 8 complex points per lane of a
// 64-lane wave; this file makes no such assumption.  One workgroup per rotation, nothing sized by a template parameter:
//   * the folded M = N/2-point transform (polynomials.jl:106-112) is an in-place mixed-radix decimation-in-frequency FFT in
//     LDS — radix-8 stages while 8 divides what is left, then one radix-4 or radix-2 stage — and leaves the spectrum in
//     digit-reversed order; the inverse (polynomials.jl:119-132) is the mirror-image decimation-in-time FFT, which consumes
//     that order.  Nothing is ever reordered: the spectrum products are element-wise, and the bootstrapping key is
//     transformed by the same code (anyn_bk_prepare_kernel) or permuted into that order (anyn_bk_permute_c128_kernel);
//   * twiddles e^{-2 pi i t/M} and the twist e^{-i pi j/N} come from tables built on the host in long double;
//   * the accumulator (k + 1 polynomials) lives in global memory (L2-resident; ordered by the workgroup barrier), the k + 1
//     spectrum accumulators of a step in LDS when they fit and in global memory otherwise (AnyNArgs::spec_g).
// Every result word is the exact negacyclic product mod 2^32 as long as the pre-rounding values stay within 1/2 of an
// integer (the DIAG instantiation measures that distance, as for every other kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "br_core.hpp"
#include "kernels_common.hpp"

using namespace tfhe;

namespace anyn {

// LDS index of complex element i: one element of padding after every 8 (the rows-of-72 / stride-9 scheme of br_core.hpp's
// transposition buffer, for any length): the stride-8 and stride-64 accesses of the later stages then spread over all banks.
__device__ __forceinline__ int phys(int i) { return i + (i >> 3); }
__host__ __device__ inline int padded_len(int M) { return M + (M >> 3); }

template <bool INV>
__device__ __forceinline__ void dft2(cplx (&x)[2])
{
    const cplx a = cadd(x[0], x[1]), b = csub(x[0], x[1]);
    x[0] = a; x[1] = b;
}
template <bool INV>
__device__ __forceinline__ void dft4(cplx (&x)[4])
{
    const cplx a0 = cadd(x[0], x[2]), a1 = cadd(x[1], x[3]), s0 = csub(x[0], x[2]), e = csub(x[1], x[3]);
    const cplx s1 = INV ? mk(-e.y, e.x) : mk(e.y, -e.x);       // e * (+-i)
    x[0] = cadd(a0, a1); x[2] = csub(a0, a1); x[1] = cadd(s0, s1); x[3] = csub(s0, s1);
}

// One stage of radix RDX on blocks of length Lb = 2^log2Lb (sub-blocks of S = Lb / RDX), all M / RDX butterflies shared out
// over the workgroup.  Forward (decimation in frequency): y_q[j] = W_Lb^{jq} * DFT_RDX(x[j + S s])[q], stored over sub-block q
// (which then holds the frequencies = q mod RDX of the block).  Inverse (decimation in time): the same in reverse with
// conjugate twiddles.  wtab[t] = e^{-2 pi i t / M}.
template <int RDX, bool INV>
__device__ __forceinline__ void stage(cplx *buf, const cplx *__restrict__ wtab, int log2M, int log2Lb, int tid, int nthreads)
{
    constexpr int LR = RDX == 8 ? 3 : RDX == 4 ? 2 : 1;
    const int log2S = log2Lb - LR, S = 1 << log2S;
    const int nb = 1 << (log2M - LR);
    const int wshift = log2M - log2Lb;                   // W_Lb^e = wtab[e << wshift]
    for (int b = tid; b < nb; b += nthreads) {
        const int blk = b >> log2S, j = b & (S - 1);
        const int base = (blk << log2Lb) + j;
        cplx x[RDX];
#pragma unroll
        for (int q = 0; q < RDX; q++) x[q] = buf[phys(base + (q << log2S))];
        if (INV) {
#pragma unroll
            for (int q = 1; q < RDX; q++) x[q] = cmulc(x[q], wtab[(j * q) << wshift]);
        }
        if constexpr (RDX == 8) dft8<INV>(x);
        else if constexpr (RDX == 4) anyn::dft4<INV>(x);
        else anyn::dft2<INV>(x);
        if (!INV) {
#pragma unroll
            for (int q = 1; q < RDX; q++) x[q] = cmul(x[q], wtab[(j * q) << wshift]);
        }
#pragma unroll
        for (int q = 0; q < RDX; q++) buf[phys(base + (q << log2S))] = x[q];
    }
    __syncthreads();
}

// forward transform of buf[0 .. M) in place (natural order in, digit-reversed out); the caller has synchronised the input
__device__ __forceinline__ void fft_fwd(cplx *buf, const cplx *__restrict__ wtab, int log2M, int tid, int nthreads)
{
    int lb = log2M;
    for (; lb >= 3; lb -= 3) stage<8, false>(buf, wtab, log2M, lb, tid, nthreads);
    if (lb == 2) stage<4, false>(buf, wtab, log2M, 2, tid, nthreads);
    else if (lb == 1) stage<2, false>(buf, wtab, log2M, 1, tid, nthreads);
}
// unnormalised inverse (the 1/M lives in the key spectra), digit-reversed in, natural order out
__device__ __forceinline__ void fft_inv(cplx *buf, const cplx *__restrict__ wtab, int log2M, int tid, int nthreads)
{
    const int rem = log2M % 3;
    if (rem == 2) stage<4, true>(buf, wtab, log2M, 2, tid, nthreads);
    else if (rem == 1) stage<2, true>(buf, wtab, log2M, 1, tid, nthreads);
    for (int lb = rem + 3; lb <= log2M; lb += 3) stage<8, true>(buf, wtab, log2M, lb, tid, nthreads);
}
// frequency held at position pos after fft_fwd: the 3-bit groups of pos (from the top) are the base-8 digits of the
// frequency from the bottom; the last, partial group is the top digit
__host__ __device__ inline int freq_of_pos(int pos, int log2M)
{
    int f = 0, weight = 0, lb = log2M;
    for (; lb >= 3; lb -= 3) { f |= ((pos >> (lb - 3)) & 7) << weight; weight += 3; }
    if (lb > 0) f |= (pos & ((1 << lb) - 1)) << weight;
    return f;
}

struct Args {
    DiagArgs diag;
    const int32_t *bara;   // [R][rows_in]   mod-switched exponents, barb last
    const cplx *bk;        // single key: [n][L][K1][K1][M]; multi-key: [P][n][2 L P + 2 L][M]; fft_fwd's order, scaled 1/M
    int32_t *ext;          // [R][(K1-1) N + 1]
    int32_t *acc;          // [R][K1][N] accumulator polynomials
    cplx *spec_g;          // [R][NSPEC][M] spectrum accumulators when they do not fit LDS, else NULL
    const cplx *wtab;      // [M]  e^{-2 pi i t / M}
    const cplx *twist;     // [M]  e^{-i pi j / N}      polynomials.jl:53
    Gadget g;
    int32_t n, mu, K1, L, R, log2N;
    int32_t parties;       // multi-key kernel only (K1 = parties + 1)
};

// temp[j] = (((X^a - 1) * poly)[j] + offset) ^ xormask for all N coefficients   (bootstrap.jl:21, tlwe.jl:88-93, tgsw.jl:104)
__device__ __forceinline__ void rotate_to_lds(const int32_t *poly, int a, int N, int32_t offset, int32_t xormask, int32_t *tmp, int tid, int nthreads)
{
    for (int j = tid; j < N; j += nthreads) {
        const int idx = (j - a) & (2 * N - 1);
        const uint32_t v = (uint32_t)poly[idx & (N - 1)];
        const uint32_t r = (idx & N) ? 0u - v : v;
        tmp[j] = (int32_t)((r - (uint32_t)poly[j] + (uint32_t)offset) ^ (uint32_t)xormask);
    }
}
// digit polynomial p (1-based) of tmp, folded and twisted into buf: z_j = (d_j - i d_{j+M}) e^{-i pi j/N}   polynomials.jl:110
__device__ __forceinline__ void digits_to_buf(const int32_t *tmp, int p, int beta, int M, const cplx *__restrict__ twist, cplx *buf, int tid, int nthreads)
{
    for (int j = tid; j < M; j += nthreads) {
        const double lo = (double)digit2(tmp[j], p, beta), hi = (double)digit2(tmp[j + M], p, beta);
        const cplx w = twist[j];
        buf[phys(j)] = mk(lo * w.x + hi * w.y, lo * w.y - hi * w.x);
    }
}
// after the inverse transform: conj(y_j) e^{-i pi j/N}: real -> coefficient j, imaginary -> j + M (polynomials.jl:127-129),
// rounded (polynomials.jl:115-116) and added into the accumulator polynomial (bootstrap.jl:22)
template <bool MARGIN>
__device__ __forceinline__ void untwist_accumulate(const cplx *buf, int M, const cplx *__restrict__ twist, int32_t *poly, double &worst, int tid, int nthreads)
{
    for (int j = tid; j < M; j += nthreads) {
        const cplx y = buf[phys(j)], w = twist[j];
        const double re = y.x * w.x + y.y * w.y, im = y.x * w.y - y.y * w.x;
        if (MARGIN) {
            const double f0 = frac_dist(re), f1 = frac_dist(im);
            worst = f0 > worst ? f0 : worst;
            worst = f1 > worst ? f1 : worst;
        }
        poly[j] = (int32_t)((uint32_t)poly[j] + (uint32_t)round_to_torus32(re));
        poly[j + M] = (int32_t)((uint32_t)poly[j + M] + (uint32_t)round_to_torus32(im));
    }
}
// element-wise out (+)= x * key over one spectrum; `out` is LDS (padded indexing) or global (linear)
__device__ __forceinline__ void mac(const cplx *buf, const cplx *__restrict__ key, cplx *out, bool out_lds, bool first, int M, int tid, int nthreads)
{
    for (int f = tid; f < M; f += nthreads) {
        const cplx x = buf[phys(f)], kv = key[f];
        cplx *o = out + (out_lds ? phys(f) : f);
        *o = first ? cmul(x, kv) : cfma(x, kv, *o);
    }
}

// LDS: buf [Mp] cplx | spectrum accumulators [nspec][Mp] cplx (only if they fit) | tmp [N] int32
__host__ __device__ inline size_t lds_bytes(int N, int nspec_in_lds)
{
    const int M = N / 2 > 0 ? N / 2 : 1;
    return (size_t)(1 + nspec_in_lds) * padded_len(M) * sizeof(cplx) + (size_t)N * sizeof(int32_t) + 16;
}
__host__ inline int threads_for(int N)
{
#ifdef TFHE_EXPERIMENT      // (measurement aid of development builds: TFHE_ANYN_THREADS overrides)
    static const char *env = getenv("TFHE_ANYN_THREADS");
    if (env && atoi(env) >= 64) return atoi(env) > 512 ? 512 : atoi(env) / 64 * 64;
#endif
    // Measured (profiles/r05/r05j_anyn_threads.txt; blind rotate of 4096 / 2048 / 1024 / 512 gates at N = 512 / 1024 / 2048 / 4096):
    //   N = 512:  64 threads 35.3 ms, 128: 30.7, 256: 38.3      N = 1024: 64: 45.9, 128: 32.1, 256: 26.7, 512: 38.3
    //   N = 2048: 128: 92.4, 256: 65.7, 512: 54.9                 N = 4096: 64: 323, 128: 176, 256: 111-113, 512: 81.4
    // A radix-8 stage has only N/16 butterflies, but the element-wise phases (rotate, digits, products, untwist) and the number of
    // waves a CU can interleave grow with the workgroup.
    return N <= 256 ? 64 : N == 512 ? 128 : N == 1024 ? 256 : 512;
}

// ---- single key (bootstrap.jl:19-82, tgsw.jl:99-129) --------------------------------------------------------------
template <bool MARGIN>
__global__ __launch_bounds__(512) void blind_rotate_kernel(Args P)
{
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = 1 << P.log2N, M = N >> 1, log2M = P.log2N - 1, Mp = padded_len(M);
    const int K1 = P.K1, L = P.L;
    const bool spec_lds = P.spec_g == nullptr;
    cplx *buf = reinterpret_cast<cplx *>(smem);
    cplx *spec = spec_lds ? buf + Mp : P.spec_g + (size_t)blockIdx.x * K1 * M;
    const int spec_stride = spec_lds ? Mp : M;
    int32_t *tmp = reinterpret_cast<int32_t *>(buf + (size_t)(spec_lds ? 1 + K1 : 1) * Mp);
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (P.n + 1);
    int32_t *acc = P.acc + w * (size_t)K1 * N;
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);

    {   // accum = (0, ..., 0, X^{-barb} (mu, ..., mu))     bootstrap.jl:54-56,78 ; tlwe.jl:77-81
        const int barb = bara[P.n] & (2 * N - 1);
        for (int e = tid; e < K1 * N; e += nt) {
            const int c = e >> P.log2N, j = e & (N - 1);
            const int idx = (j + barb) & (2 * N - 1);
            acc[e] = c + 1 < K1 ? 0 : (idx & N) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int i = 0; i < P.n; i++) {                                      // bootstrap.jl:32-39 (a zero exponent adds exactly zero)
        const int a = bara[i] & (2 * N - 1);
        const cplx *key = P.bk + (size_t)i * ((size_t)L * K1 * K1 * M);
#pragma unroll 1
        for (int c = 0; c < K1; c++) {
            rotate_to_lds(acc + (size_t)c * N, a, N, P.g.offset, xormask, tmp, tid, nt);
            __syncthreads();
#pragma unroll 1
            for (int p = 0; p < L; p++) {
                digits_to_buf(tmp, p + 1, beta, M, P.twist, buf, tid, nt);
                __syncthreads();
                fft_fwd(buf, P.wtab, log2M, tid, nt);
                // out[co] += D[p, c] .* BK_i[p, c].a[co]        tgsw.jl:128
#pragma unroll 1
                for (int co = 0; co < K1; co++)
                    mac(buf, key + (size_t)((p * K1 + c) * K1 + co) * M, spec + (size_t)co * spec_stride, spec_lds, c == 0 && p == 0, M, tid, nt);
                __syncthreads();
            }
        }
        // every rotated read of this step is done: inverse transforms, rounding, accumulator update (bootstrap.jl:22)
#pragma unroll 1
        for (int co = 0; co < K1; co++) {
            cplx *y = spec + (size_t)co * spec_stride;
            if (!spec_lds) {
                for (int f = tid; f < M; f += nt) buf[phys(f)] = y[f];
                __syncthreads();
                y = buf;
            }
            fft_inv(y, P.wtab, log2M, tid, nt);
            untwist_accumulate<MARGIN>(y, M, P.twist, acc + (size_t)co * N, worst, tid, nt);
            __syncthreads();
        }
    }

    // tlwe_extract_sample (tlwe.jl:55-59): a'[0] = p[0], a'[m] = -p[N - m], mask polynomials in order; b = body[0]
    int32_t *ext = P.ext + w * ((size_t)(K1 - 1) * N + 1);
    for (int e = tid; e < (K1 - 1) * N; e += nt) {
        const int c = e >> P.log2N, j = e & (N - 1);
        const int32_t v = acc[e];
        if (j == 0) ext[(size_t)c * N] = v;
        else ext[(size_t)c * N + N - j] = (int32_t)(0u - (uint32_t)v);
    }
    if (tid == 0) ext[(size_t)(K1 - 1) * N] = acc[(size_t)(K1 - 1) * N];
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
}

// ---- multi-key (mk_internals.jl:348-391,464-495), any number of parties ---------------------------------------------
// Step (party i, bit j) with digits da[p, s] of mask s and db[p] of the body (spectrum-domain sums, as every MK kernel here):
//   a'_s     += sum_p da[p, s] (*) y[p, i]                            for every s != i        (:377-378)
//   a'_i     += sum_{p, s} da[p, s] (*) y[p, s] + sum_p db[p] (*) c1[p]                        (:371-376)
//   b'       += sum_{p, s} da[p, s] (*) x[p, s] + sum_p db[p] (*) c0[p]                        (:382-385)
// Three spectrum accumulators are live whatever P is: a'_s of a non-party source is fed by its own digits only and nobody
// else reads acc[s] in this step, so it is finished right after source s.
template <bool MARGIN>
__global__ __launch_bounds__(512) void mk_blind_rotate_kernel(Args P)
{
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = 1 << P.log2N, M = N >> 1, log2M = P.log2N - 1, Mp = padded_len(M);
    const int NP = P.parties, L = P.L;
    const bool spec_lds = P.spec_g == nullptr;
    cplx *buf = reinterpret_cast<cplx *>(smem);
    cplx *spec = spec_lds ? buf + Mp : P.spec_g + (size_t)blockIdx.x * 3 * M;      // [self | party | body]
    const int ss = spec_lds ? Mp : M;
    int32_t *tmp = reinterpret_cast<int32_t *>(buf + (size_t)(spec_lds ? 4 : 1) * Mp);
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * ((size_t)NP * P.n + 1);
    int32_t *acc = P.acc + w * (size_t)(NP + 1) * N;
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);
    const int per = 2 * L * NP + 2 * L;

    {
        const int barb = bara[(size_t)NP * P.n] & (2 * N - 1);
        for (int e = tid; e < (NP + 1) * N; e += nt) {
            const int c = e >> P.log2N, j = e & (N - 1);
            const int idx = (j + barb) & (2 * N - 1);
            acc[e] = c < NP ? 0 : (idx & N) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
    }
    __syncthreads();

    auto finish = [&](int which, int d) {      // inverse transform of spectrum accumulator `which`, round, add into polynomial d
        cplx *y = spec + (size_t)which * ss;
        if (!spec_lds) {
            for (int f = tid; f < M; f += nt) buf[phys(f)] = y[f];
            __syncthreads();
            y = buf;
        }
        fft_inv(y, P.wtab, log2M, tid, nt);
        untwist_accumulate<MARGIN>(y, M, P.twist, acc + (size_t)d * N, worst, tid, nt);
        __syncthreads();
    };

#pragma unroll 1
    for (int party = 0; party < NP; party++) {                               // mk_internals.jl:475
#pragma unroll 1
        for (int j = 0; j < P.n; j++) {                                      // :476
            const int a = bara[(size_t)party * P.n + j] & (2 * N - 1);
            const cplx *key = P.bk + ((size_t)party * P.n + j) * per * M;
            bool first_pb = true;
#pragma unroll 1
            for (int s = 0; s <= NP; s++) {
                const bool is_body = (s == NP), has_self = (!is_body && s != party);
                rotate_to_lds(acc + (size_t)s * N, a, N, P.g.offset, xormask, tmp, tid, nt);
                __syncthreads();
#pragma unroll 1
                for (int p = 0; p < L; p++) {
                    digits_to_buf(tmp, p + 1, beta, M, P.twist, buf, tid, nt);
                    __syncthreads();
                    fft_fwd(buf, P.wtab, log2M, tid, nt);
                    const cplx *k_party = key + (size_t)(is_body ? 2 * L * NP + L + p : L * NP + p * NP + s) * M;   // c1[p] | y[p, s]
                    const cplx *k_body = key + (size_t)(is_body ? 2 * L * NP + p : p * NP + s) * M;                 // c0[p] | x[p, s]
                    mac(buf, k_party, spec + (size_t)1 * ss, spec_lds, first_pb, M, tid, nt);
                    mac(buf, k_body, spec + (size_t)2 * ss, spec_lds, first_pb, M, tid, nt);
                    first_pb = false;
                    if (has_self) mac(buf, key + (size_t)(L * NP + p * NP + party) * M, spec, spec_lds, p == 0, M, tid, nt);   // y[p, party]
                    __syncthreads();
                }
                if (has_self) finish(0, s);
            }
            finish(1, party);
            finish(2, NP);
        }
    }

    // mk_tlwe_extract_sample (mk_internals.jl:88-95): per party a'[0] = p[0], a'[m] = -p[N - m]; b = body[0]
    int32_t *ext = P.ext + w * ((size_t)NP * N + 1);
    for (int e = tid; e < NP * N; e += nt) {
        const int c = e >> P.log2N, j = e & (N - 1);
        const int32_t v = acc[e];
        if (j == 0) ext[(size_t)c * N] = v;
        else ext[(size_t)c * N + N - j] = (int32_t)(0u - (uint32_t)v);
    }
    if (tid == 0) ext[(size_t)NP * N] = acc[(size_t)NP * N];
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
}

#ifdef TFHE_EMIT_KEYPREP_KERNELS       // (defined by engine_keys.hip, the one translation unit that launches them)
// ---- bootstrapping-key preparation --------------------------------------------------------------------------------
// Int32 polynomial -> spectrum in fft_fwd's order, scaled (1/M for key polynomials: forward_transform.(bk), bootstrap.jl:12)
__global__ __launch_bounds__(512) void bk_prepare_kernel(const int32_t *__restrict__ polys, cplx *__restrict__ out, const cplx *__restrict__ wtab,
                                                         const cplx *__restrict__ twist, int log2N, double scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *buf = reinterpret_cast<cplx *>(smem);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = 1 << log2N, M = N >> 1;
    const size_t q = blockIdx.x;
    const int32_t *poly = polys + q * N;
    for (int j = tid; j < M; j += nt) {
        const double lo = (double)poly[j], hi = (double)poly[j + M];
        const cplx w = twist[j];
        buf[phys(j)] = mk(lo * w.x + hi * w.y, lo * w.y - hi * w.x);
    }
    __syncthreads();
    fft_fwd(buf, wtab, log2N - 1, tid, nt);
    for (int f = tid; f < M; f += nt) {
        const cplx v = buf[phys(f)];
        out[q * M + f] = mk(v.x * scale, v.y * scale);
    }
}
// The reference's stored spectra (natural frequency order, polynomials.jl:106-112) -> fft_fwd's order, scaled 1/M
__global__ void bk_permute_c128_kernel(const cplx *__restrict__ in, cplx *__restrict__ out, int log2M, size_t total)
{
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int M = 1 << log2M;
    const size_t q = e >> log2M;
    const int pos = (int)(e & (size_t)(M - 1));
    const cplx v = in[q * M + freq_of_pos(pos, log2M)];
    const double s = 1.0 / (double)M;
    out[e] = mk(v.x * s, v.y * s);
}

// ---- RGSW.Expand (mk_internals.jl:304-345) for any N: the counterpart of mk_expand_kernel (kernels_keyprep.hpp) ----------
//     x[jj, q] = d0[jj] + sum_u g^-1(b_q[jj] - b_i[jj])[u] (*) f0[u]          y[jj, q] = sum_u g^-1(...)[u] (*) f1[u]
// One workgroup per output polynomial: l spectrum products, one inverse transform, one rounding.
#endif  // TFHE_EMIT_KEYPREP_KERNELS
struct MkExpandArgs {
    const cplx *dec;      // [P-1 (other party, in order)][l (u)][l (jj)][M]   digit spectra, unscaled
    const cplx *f;        // [2 (f0 | f1)][n][l (u)][M]                          spectra scaled 1/M
    const int32_t *d0;    // [n][l][N]
    int32_t *key;         // [n][2 l P + 2 l][N]   (x and y slots of the OTHER parties are written here)
    const cplx *wtab, *twist;
    int32_t n, l, parties, party, log2N;
};
#ifdef TFHE_EMIT_KEYPREP_KERNELS
__global__ __launch_bounds__(512) void mk_expand_kernel(MkExpandArgs A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *buf = reinterpret_cast<cplx *>(smem);
    const int tid = threadIdx.x, nt = blockDim.x;
    const int N = 1 << A.log2N, M = N >> 1;
    // grid: x = bit j, y = (jj, other-party index oq), z = 0 (x) | 1 (y)
    const int j = blockIdx.x, jj = blockIdx.y % A.l, oq = blockIdx.y / A.l, xy = blockIdx.z;
    const int q = oq < A.party ? oq : oq + 1;                        // the oq-th party other than `party`
    const int per = 2 * A.l * A.parties + 2 * A.l;
    for (int f = tid; f < M; f += nt) {
        cplx acc = mk(0.0, 0.0);
        for (int u = 0; u < A.l; u++)
            acc = cfma(A.dec[((size_t)(oq * A.l + u) * A.l + jj) * M + f], A.f[(((size_t)xy * A.n + j) * A.l + u) * M + f], acc);
        buf[phys(f)] = acc;
    }
    __syncthreads();
    fft_inv(buf, A.wtab, A.log2N - 1, tid, nt);
    int32_t *o = A.key + ((size_t)j * per + (xy == 0 ? 0 : A.l * A.parties) + jj * A.parties + q) * N;
    const int32_t *d0 = A.d0 + ((size_t)j * A.l + jj) * N;
    for (int t = tid; t < M; t += nt) {
        const cplx y = buf[phys(t)], w = A.twist[t];
        const double re = y.x * w.x + y.y * w.y, im = y.x * w.y - y.y * w.x;
        // y has no d1 term for q != party (:334-339)
        o[t] = (int32_t)((xy == 0 ? (uint32_t)d0[t] : 0u) + (uint32_t)round_to_torus32(re));
        o[t + M] = (int32_t)((xy == 0 ? (uint32_t)d0[t + M] : 0u) + (uint32_t)round_to_torus32(im));
    }
}
#endif  // TFHE_EMIT_KEYPREP_KERNELS

}  // namespace anyn
