// kernels_w2.hpp — blind_rotate_kernel_w2: two waves per blind rotation (257 … 1024 rotations and the tails of split launches).
#pragma once
#include "kernels_common.hpp"

// ---- small and medium batches (up to 1024 rotations): two waves per blind rotation --------------------
// With fewer rotations than wave slots (single gates, sequential circuits, small batches) one wave per
// rotation leaves the chip idle and a gate takes n x (4 forward + 2 inverse transforms) of latency.
// Here wave c (c = 0: mask polynomial, c = 1: body) owns accumulator polynomial c: it rotates and
// decomposes only its own polynomial, runs its L forward transforms, MACs both output components, hands
// the partial sum for the other component over through LDS (the two transposition buffers change hands every
// step: ONE barrier per step, no separate hand-off area), adds what it receives, inverse-transforms its own
// component and updates its own polynomial.  Same arithmetic per rotation as blind_rotate_kernel_v3, about
// half the latency; 27.4 KB of LDS and < 256 registers per wave, so 1024 rotations are resident at two waves
// per SIMD (3.6 ms for 1024 rotations against 5.8 ms with one wave per rotation and SIMD).
// (Measured dead end: one wave per (component, digit) — 2 l waves, one forward transform each — is no faster, 1.94 vs
//  1.89 ms per gate: a lone wave issues FP64 at about half the SIMD's rate, and four waves transposing at once run into
//  the CU's LDS store bandwidth, so every transform gets slower as the step gets shorter.)
constexpr int kW2LdsBytes = 2 * kImg * 4 + (2 * kXchElems + 64) * (int)sizeof(cplx);     // per rotation
// RW rotations per workgroup (RW = 2: the step barrier then spans both rotations, which keeps them in lockstep and lets
// them share their key reads in the CU's L1, as in the other kernels; a padding rotation repeats the last one and stores nothing)
template <int L, bool MARGIN = false, int RW = 1>
__global__ __launch_bounds__(128 * RW, 2) void blind_rotate_kernel_w2(BrArgs P)
{
    constexpr int K1 = 2;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int wib = wave_in_block();
    char *smem = smem_all + (size_t)(RW > 1 ? (wib >> 1) : 0) * kW2LdsBytes;
    int32_t *acc_all = reinterpret_cast<int32_t *>(smem);                        // [K1][kImg]
    cplx *xch_all = reinterpret_cast<cplx *>(smem + K1 * kImg * 4);              // [2][kXchElems]: the waves swap them every step
    cplx *tw2_lds = xch_all + 2 * kXchElems;                                     // [8][8]
    const int tid = threadIdx.x & 127, lane = tid & 63;
    const int wv = RW > 1 ? (wib & 1) : wib;                                      // wave = owned polynomial
    int32_t *acc_lds = acc_all + wv * kImg;
    size_t w = (size_t)blockIdx.x * RW + (RW > 1 ? (wib >> 1) : 0);
    const bool padding = RW > 1 && w >= (size_t)P.R;
    if (padding) w = (size_t)P.R - 1;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int Lr = L ? L : P.l;                   // L = 0: any decomposition length at run time (see blind_rotate_kernel_v3)
    const int32_t xormask = gadget_xor_mask(Lr, beta);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    if (tid < 64) tw2_lds[tid] = P.T.tw2[tid];
    if (wv) init_body_poly(lane, bara[P.n] & (2 * kN - 1), P.mu, acc_lds);
    else init_zero_poly(lane, acc_lds);
    __syncthreads();
    STAMP_DECL;

    int a_next = load_uniform_i32(bara) & (2 * kN - 1);
    wave_priority_begin(P.prio_steps);
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        wave_priority_step(i, P.prio_steps);
        const int a = a_next;
        a_next = load_uniform_i32(bara + i + 1) & (2 * kN - 1);   // bara[n] (= barb) exists: harmless read on the last step
        // key polys of transform (p, c = wv): [i][p][c][co][8][64]
        const cplx *key = P.bk + (size_t)i * (Lr * K1 * K1 * kM) + (size_t)wv * K1 * kM + lane;
        // Transposition buffers: in step i this wave transforms in buffer (wv ^ i) & 1 and leaves its hand-off there; after
        // the barrier it reads the other wave's hand-off from the other buffer and runs its inverse transform in it — and
        // keeps that buffer for the forward transforms of step i + 1, while the other wave has moved to this one.  One
        // barrier per step, no separate hand-off area (27.4 KB of LDS per rotation).
        cplx *xch = xch_all + ((wv ^ i) & 1) * kXchElems, *xch_next = xch_all + ((wv ^ i ^ 1) & 1) * kXchElems;
        cplx own[8], oth[8];
        int32_t temp[16];
        rotate_poly<16>(lane, a, acc_lds, P.g.offset, xormask, temp);
        STAMP(0);
        // digit p: transform, multiply into both output components (the first digit's products are written, not accumulated:
        // nothing to zero — as in blind_rotate_kernel_v3)
        auto digit = [&](int p, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            cplx x[8];
            load_digits2t(temp, p + 1, beta, x);
            const cplx *kp = key + (size_t)p * K1 * K1 * kM;
            cplx kown[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) kown[k2] = kp[(size_t)wv * kM + k2 * 64];           // co = wv (issued before the FFT)
            fft_fwd_wave<true>(lane, x, tw1f, tw2_lds, xch);
            STAMP(1);
            cplx koth[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) koth[k2] = kp[(size_t)(1 - wv) * kM + k2 * 64];     // co = 1 - wv
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) own[k2] = FIRST ? cmul(x[k2], kown[k2]) : cfma(x[k2], kown[k2], own[k2]);
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) oth[k2] = FIRST ? cmul(x[k2], koth[k2]) : cfma(x[k2], koth[k2], oth[k2]);
            STAMP(2);
        };
        digit(0, std::true_type{});
#pragma unroll 1
        for (int p = 1; p < Lr; p++) digit(p, std::false_type{});
        // hand the other component's partial sum over
        WAVE_LDS_FENCE();
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) xch[k2 * 64 + lane] = oth[k2];
        STAMP(3);
        __syncthreads();
        STAMP(4);
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) own[k2] = cadd(own[k2], xch_next[k2 * 64 + lane]);
        WAVE_LDS_FENCE();
        STAMP(5);
        fft_inv_wave(lane, own, tw1f, tw2_lds, xch_next);
        STAMP(6);
        accumulate_poly<MARGIN>(lane, own, acc_lds, &worst);
        WAVE_LDS_FENCE();
        STAMP(7);
    }
    STAMP_FLUSH(P.diag, wv);
    __syncthreads();
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
    if (padding) return;
    int32_t *ext = P.ext + w * (kN + 1);
    if (wv == 0) extract_mask_poly(lane, acc_all, ext);
    else if (lane == 0) ext[kN] = acc_all[kImg + kMir];
}
