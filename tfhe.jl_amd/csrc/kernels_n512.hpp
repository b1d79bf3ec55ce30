// kernels_n512.hpp — tuned blind rotation for N = 512, k = 1 (round 5).
//
// The reference ships N = 1024 sets only, but SchemeParameters takes any degree (api.jl:4-21) and N = 512 is the usual small
// parameter choice of gate-bootstrapping TFHE.  The any-N kernel (kernels_anyn.hpp) runs such a set through LDS-resident generic
// transforms; this kernel is blind_rotate_kernel_v3's design at half the size: one 64-lane wave owns one accumulator for all n
// CMUX steps, a polynomial of 512 coefficients is M = 256 folded complex points, FOUR per lane (lane t holds points t + 64 r,
// r < 4, i.e. coefficients t + 64 m, m < 8), and the 256-point transform is the four-radix-4-pass, three-transposition transform
// blind_rotate_kernel_h2 already uses for its half-transforms (fft256_fwd / fft256_inv, kernels_h2.hpp) with the lane
// part of THIS degree's twist folded into its first twiddles.  With 16 registers of transform state instead of 32 the wave fits
// three to a SIMD (152 registers, no scratch); 9.7 KB of LDS per rotation.  (Measured and not kept: FOUR waves per SIMD — 119
// registers with the later passes' twiddles in a 1.3 KB LDS table per workgroup, four workgroups of four rotations exactly filling a
// CU's LDS — is no faster at 4096 rotations (7.26 - 7.59 vs 7.23 ms), slower at 8192 (13.6 - 14.2 vs 12.9) and for a single rotation
// (2.88 vs 2.03 ms): three transpositions per 256-point transform already keep the LDS pipe busy, and six more table reads per
// transform cost more than the fourth wave brings.)
// Spectrum order: lane (q, q2, q3) = 16 q + 4 q2 + q3, register q4 <-> frequency
// q + 4 q2 + 16 q3 + 64 q4, which the inverse consumes and the key is stored in ([n][l][2][2][4 (q4)][64 (lane)], scaled 1/M).
#pragma once
#include <hip/hip_runtime.h>

#include "br_core.hpp"
#include "kernels_h2.hpp"      // the 256-point transforms (fft256_fwd / fft256_inv)

using namespace tfhe;

constexpr int kN5 = 512;
constexpr int kM5 = kN5 / 2;
constexpr int kImg5 = kMir + kN5;            // one polynomial in LDS: mirror | coefficients (rotate_sub3<8>)
constexpr int kN512TableElems = 4 * 64;      // tw1 [4 (q)][64 (t)] = e^{-i pi t/512} e^{-2 pi i t q/256}; the later passes share blind_rotate_kernel_h2's tables
constexpr int kN512LdsBytes = 2 * kImg5 * 4 + kH2Buf * (int)sizeof(cplx);      // per rotation

struct N512Args {
    DiagArgs diag;
    const int32_t *bara;   // [R][n+1], barb last
    const cplx *bk;        // [n][L][2][2][4][64]
    int32_t *ext;          // [R][N+1]
    const cplx *tw1;       // [4][64]
    const cplx *tw2q;      // [4][16]  e^{-2 pi i t1 q2/64}
    const cplx *tw3q;      // [4][4]   e^{-2 pi i t2 q3/16}
    Gadget g;
    int32_t n, mu, R, l;
    int32_t prio_steps;
};

// u[r] = (d[t + 64 r] - i d[t + 64 r + 256]) e^{-i pi r/8} / twk(2 r): the register part of the twist e^{-i pi (t + 64 r)/512}
// (polynomials.jl:110) in tan form (br_core.hpp, twist_tan); the cosines (1, c2, c4, c2) ride on the first butterfly (kN512Scale)
__device__ __forceinline__ void load_digits4t(const int32_t (&temp)[8], int p, int log2_base, cplx (&u)[4])
{
#define TFHE_TW_(R) u[R] = twist_tan<2 * R>((double)digit2(temp[R], p, log2_base), (double)digit2(temp[R + 4], p, log2_base))
    TFHE_TW_(0); TFHE_TW_(1); TFHE_TW_(2); TFHE_TW_(3);
#undef TFHE_TW_
}
#define kN512Scale Dft4Scale{kTwR0, 1.0, kTwG0}      // S = (1, c2, c4, c2): g0 = c4, g1 = c2 / c2, s1 = c2
// conj(y) e^{-i pi r/8}: real -> coefficient t + 64 r, imaginary -> t + 64 r + 256; round, add   polynomials.jl:115-116,127-129
// (tan form; the cosine rides on the FMA that adds the rounding constant: untwist_add2)
template <bool MARGIN>
__device__ __forceinline__ void untwist_add4(const cplx (&y)[4], int32_t (&acc)[8], double &worst)
{
#pragma unroll
    for (int r = 0; r < 4; r++) {
        double zr, zi;
        if (r == 0) { zr = y[r].x; zi = y[r].y; }
        else if (r == 2) { zr = y[r].x - y[r].y; zi = y[r].y + y[r].x; }
        else if (r == 1) { zr = fma_(-twt(2), y[r].y, y[r].x); zi = fma_(twt(2), y[r].x, y[r].y); }
        else { zr = fma_(twt(6), y[r].x, -y[r].y); zi = fma_(twt(6), y[r].y, y[r].x); }
        if (MARGIN) {
            const double a = frac_dist(zr * twk(2 * r)), b = frac_dist(zi * twk(2 * r));
            worst = a > worst ? a : worst;
            worst = b > worst ? b : worst;
        }
        acc[r] = (int32_t)((uint32_t)acc[r] + (uint32_t)round_scaled_to_torus32(zr, twk(2 * r)));
        acc[r + 4] = (int32_t)((uint32_t)acc[r + 4] + (uint32_t)round_scaled_to_torus32(zi, -twk(2 * r)));
    }
}

// L = 0: the decomposition length is a run-time value (P.l), as in blind_rotate_kernel_v3<0, ...>.  RW rotations per workgroup in
// lockstep (one barrier every kV3SyncEvery steps: they stream the same key lines together), a padding wave repeats the last rotation.
template <int L, bool MARGIN = false, int RW = 1>
__global__ __launch_bounds__(64 * RW, 3) void blind_rotate_kernel_n512(N512Args P)
{
    constexpr int K1 = 2;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int wib = (RW > 1) ? wave_in_block() : 0;
    char *smem = smem_all + (size_t)wib * kN512LdsBytes;
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem);                    // [K1][kImg5]
    cplx *tb = reinterpret_cast<cplx *>(smem + K1 * kImg5 * 4);              // [kH2Buf]
    const int lane = (RW > 1) ? lane_id() : (int)threadIdx.x;
    size_t w = (size_t)blockIdx.x * RW + wib;
    const bool padding = (RW > 1) && w >= (size_t)P.R;
    if (padding) w = (size_t)P.R - 1;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int Lr = L ? L : P.l;
    const int32_t xormask = gadget_xor_mask(Lr, beta);
    wave_priority_begin(P.prio_steps);

    H2LaneTw tw;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        tw.tw1[q] = P.tw1[q * 64 + lane];
        tw.tw2[q] = P.tw2q[q * 16 + (lane & 15)];
        tw.tw3[q] = P.tw3q[q * 4 + (lane & 3)];
    }
    {   // accum = (0, X^{-barb} (mu, ..., mu))     bootstrap.jl:54-56,78 ; tlwe.jl:77-81
        const int barb = bara[P.n] & (2 * kN5 - 1);
        int32_t v[8];
#pragma unroll
        for (int m = 0; m < 8; m++) v[m] = 0;
        store_cur<8>(lane, v, acc_lds);
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int idx = (lane + 64 * m + barb) & (2 * kN5 - 1);
            v[m] = (idx & kN5) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
        store_cur<8>(lane, v, acc_lds + kImg5);
    }
    WAVE_LDS_FENCE();

    int a_next = load_uniform_i32(bara) & (2 * kN5 - 1);
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        wave_priority_step(i, P.prio_steps);
        const int a = a_next;
        a_next = load_uniform_i32(bara + i + 1) & (2 * kN5 - 1);   // bara[n] (= barb) exists: harmless read on the last step
        if (RW > 1 && (i % kV3SyncEvery) == 0) __builtin_amdgcn_s_barrier();
        const cplx *key = P.bk + (size_t)i * (Lr * K1 * K1 * kM5) + lane;          // [p][c][co][4][64]
        cplx out[K1][4];
#pragma unroll 1
        for (int c = 0; c < K1; c++) {
            int32_t temp[8];
            rotate_poly<8>(lane, a, acc_lds + c * kImg5, P.g.offset, xormask, temp);
#pragma unroll 1
            for (int p = 0; p < Lr; p++) {
                const cplx *kp = key + (size_t)(p * K1 + c) * K1 * kM5;
                cplx k0[4], k1[4];
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) { k0[q4] = kp[q4 * 64]; k1[q4] = kp[kM5 + q4 * 64]; }       // requested before the transform
                cplx x[4];
                load_digits4t(temp, p + 1, beta, x);
                fft256_fwd<true>(lane, x, tw, tb, kN512Scale);
                // out[co] (+)= D[p, c] .* BK_i[p, c].a[co]   (tgsw.jl:128); the step's first transform writes (nothing to zero)
                if (c == 0 && p == 0) {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) { out[0][q4] = cmul(x[q4], k0[q4]); out[1][q4] = cmul(x[q4], k1[q4]); }
                } else {
#pragma unroll
                    for (int q4 = 0; q4 < 4; q4++) { out[0][q4] = cfma(x[q4], k0[q4], out[0][q4]); out[1][q4] = cfma(x[q4], k1[q4], out[1][q4]); }
                }
            }
        }
#pragma unroll
        for (int co = 0; co < K1; co++) {
            fft256_inv(lane, out[co], tw, tb);
            int32_t accr[8];
            load_cur<8>(lane, acc_lds + co * kImg5, accr);
            untwist_add4<MARGIN>(out[co], accr, worst);
            store_cur<8>(lane, accr, acc_lds + co * kImg5);
        }
        WAVE_LDS_FENCE();
    }
    if (padding) return;
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0, lane == 0);
    // tlwe_extract_sample (tlwe.jl:55-59): a'[0] = p[0], a'[m] = -p[N - m]; b = body[0]
    int32_t *ext = P.ext + w * (kN5 + 1);
#pragma unroll
    for (int m = 0; m < 8; m++) {
        const int j = lane + 64 * m;
        const int32_t v = acc_lds[kMir + j];
        if (j == 0) ext[0] = v;
        else ext[kN5 - j] = (int32_t)(0u - (uint32_t)v);
    }
    if (lane == 0) ext[kN5] = acc_lds[kImg5 + kMir];
}

#ifdef TFHE_EMIT_KEYPREP_KERNELS       // (defined by engine_keys.hip, the one translation unit that launches them)
// key preparation: Int32 polynomial -> spectrum in the kernel's order, scaled 1/M (forward_transform.(bk), bootstrap.jl:12)
__global__ __launch_bounds__(64) void bk_prepare_kernel_n512(const int32_t *__restrict__ polys, cplx *__restrict__ out, const cplx *__restrict__ tw1,
                                                            const cplx *__restrict__ tw2q, const cplx *__restrict__ tw3q)
{
    __shared__ __attribute__((aligned(16))) cplx tb[kH2Buf];
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    const int32_t *poly = polys + q * kN5;
    H2LaneTw tw;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        tw.tw1[k] = tw1[k * 64 + lane];
        tw.tw2[k] = tw2q[k * 16 + (lane & 15)];
        tw.tw3[k] = tw3q[k * 4 + (lane & 3)];
    }
    cplx x[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const double lo = (double)poly[lane + 64 * r], hi = (double)poly[lane + 64 * r + kM5];
        if (r == 0) x[r] = mk(lo, -hi);
        else x[r] = mk(lo * twc(2 * r) - hi * tws(2 * r), -(lo * tws(2 * r) + hi * twc(2 * r)));
    }
    fft256_fwd(lane, x, tw, tb);
    const double s = 1.0 / kM5;
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) out[q * kM5 + q4 * 64 + lane] = mk(x[q4].x * s, x[q4].y * s);
}

// the reference's stored spectra (natural frequency order, polynomials.jl:106-112) -> the kernel's order, scaled 1/M
__global__ __launch_bounds__(64) void bk_permute_c128_kernel_n512(const cplx *__restrict__ in, cplx *__restrict__ out)
{
    const int lane = threadIdx.x;
    const size_t q = blockIdx.x;
    const int k0 = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);
    const double s = 1.0 / kM5;
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
        const cplx v = in[q * kM5 + k0 + 64 * q4];
        out[q * kM5 + q4 * 64 + lane] = mk(v.x * s, v.y * s);
    }
}
#endif  // TFHE_EMIT_KEYPREP_KERNELS

// ---- N = 512, small and medium batches: TWO waves per rotation (blind_rotate_kernel_w2's structure at this degree) ------------------
// One wave per rotation leaves a CU's SIMDs idle below 12 rotations per CU and makes a single gate 500 x (4 forward + 2 inverse
// transforms) of one wave's latency long (1.96 ms — longer than a single N = 1024 gate on blind_rotate_kernel_h2).  Here wave c owns
// accumulator polynomial c: it rotates and decomposes only its own polynomial, runs its L forward transforms, multiplies into
// partial sums of both output components, hands the other component's over through its transposition buffer (the two buffers
// change hands every step: ONE barrier per step), adds what it receives, inverse-transforms its own component and updates its
// own polynomial.  14.8 KB of LDS per rotation.
constexpr int kN512W2LdsBytes = 2 * kImg5 * 4 + 2 * kH2Buf * (int)sizeof(cplx);      // per rotation
template <int L, bool MARGIN = false>
__global__ __launch_bounds__(128, 3) void blind_rotate_kernel_n512w2(N512Args P)
{
    constexpr int K1 = 2;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_all = reinterpret_cast<int32_t *>(smem);                        // [K1][kImg5]
    cplx *tb_all = reinterpret_cast<cplx *>(smem + K1 * kImg5 * 4);              // [2][kH2Buf]: the waves swap them every step
    const int wv = wave_in_block();                                              // wave = owned polynomial
    const int lane = (int)threadIdx.x & 63;
    int32_t *acc_lds = acc_all + wv * kImg5;
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int Lr = L ? L : P.l;
    const int32_t xormask = gadget_xor_mask(Lr, beta);

    H2LaneTw tw;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        tw.tw1[q] = P.tw1[q * 64 + lane];
        tw.tw2[q] = P.tw2q[q * 16 + (lane & 15)];
        tw.tw3[q] = P.tw3q[q * 4 + (lane & 3)];
    }
    {
        const int barb = bara[P.n] & (2 * kN5 - 1);
        int32_t v[8];
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int idx = (lane + 64 * m + barb) & (2 * kN5 - 1);
            v[m] = wv == 0 ? 0 : (idx & kN5) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
        store_cur<8>(lane, v, acc_lds);
    }
    __syncthreads();

    int a_next = load_uniform_i32(bara) & (2 * kN5 - 1);
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        const int a = a_next;
        a_next = load_uniform_i32(bara + i + 1) & (2 * kN5 - 1);
        // key polys of transform (p, c = wv): [i][p][c][co][4][64]
        const cplx *key = P.bk + (size_t)i * (Lr * K1 * K1 * kM5) + (size_t)wv * K1 * kM5 + lane;
        // in step i this wave transforms in buffer (wv ^ i) & 1 and leaves its hand-off there; after the barrier it reads the other
        // wave's hand-off from the other buffer, runs its inverse transform in it and keeps it for the next step's forward transforms
        cplx *tb = tb_all + ((wv ^ i) & 1) * kH2Buf, *tb_next = tb_all + ((wv ^ i ^ 1) & 1) * kH2Buf;
        cplx own[4], oth[4];
        int32_t temp[8];
        rotate_poly<8>(lane, a, acc_lds, P.g.offset, xormask, temp);
#pragma unroll 1
        for (int p = 0; p < Lr; p++) {
            const cplx *kp = key + (size_t)p * K1 * K1 * kM5;
            cplx kown[4], koth[4];
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) { kown[q4] = kp[(size_t)wv * kM5 + q4 * 64]; koth[q4] = kp[(size_t)(1 - wv) * kM5 + q4 * 64]; }
            cplx x[4];
            load_digits4t(temp, p + 1, beta, x);
            fft256_fwd<true>(lane, x, tw, tb, kN512Scale);
            if (p == 0) {
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) { own[q4] = cmul(x[q4], kown[q4]); oth[q4] = cmul(x[q4], koth[q4]); }
            } else {
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) { own[q4] = cfma(x[q4], kown[q4], own[q4]); oth[q4] = cfma(x[q4], koth[q4], oth[q4]); }
            }
        }
        WAVE_LDS_FENCE();
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) tb[q4 * 64 + lane] = oth[q4];
        __syncthreads();
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) own[q4] = cadd(own[q4], tb_next[q4 * 64 + lane]);
        WAVE_LDS_FENCE();
        fft256_inv(lane, own, tw, tb_next);
        int32_t accr[8];
        load_cur<8>(lane, acc_lds, accr);
        untwist_add4<MARGIN>(own, accr, worst);
        store_cur<8>(lane, accr, acc_lds);
        WAVE_LDS_FENCE();
    }
    __syncthreads();
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
    int32_t *ext = P.ext + w * (kN5 + 1);
    if (wv == 0) {
#pragma unroll
        for (int m = 0; m < 8; m++) {
            const int j = lane + 64 * m;
            const int32_t v = acc_all[kMir + j];
            if (j == 0) ext[0] = v;
            else ext[kN5 - j] = (int32_t)(0u - (uint32_t)v);
        }
    } else if (lane == 0) {
        ext[kN5] = acc_all[kImg5 + kMir];
    }
}
