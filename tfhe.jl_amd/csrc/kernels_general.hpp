// kernels_general.hpp — blind_rotate_kernel_general: any single-key set at N = 1024 / 2048 the specialised kernels leave (k <= 4, any l).
#pragma once
#include "kernels_common.hpp"
#include "kernels_n2048.hpp"

// ---- any single-key parameter set (round 4): run-time mask size k <= 4 and decomposition length l, N = 1024 or 2048 ----
// SchemeParameters is an unvalidated struct and tlwe_mask_size a free keyword in the reference (api.jl:4-21,30,55): a
// parameter set it accepts must not be refused here because no specialised kernel was instantiated for it.  This kernel
// takes what the others leave (k >= 3, l >= 5, N = 2048 with k >= 2): one wave per rotation, one wave per SIMD, nothing
// tuned.  The accumulator images (k + 1 polynomials) live in global memory (L2-resident; a wave reads back only what it
// wrote itself, ordered by a workgroup-scope fence per step, as in mk_blind_rotate_kernel_general's ACCG variant), the
// k + 1 spectrum accumulators of a step in LDS ((k + 1) x N/1024 x 8 KB), so no register array depends on k or l.
// N = 2048: the radix-2 split of blind_rotate_kernel_n2048x with both halves computed by the one wave, one after the other.
struct BrGenArgs {
    DiagArgs diag;
    const int32_t *bara;  // [R][n+1]
    const cplx *bk;       // [n][L][K1][K1][H][8][64], H = N / 1024 halves
    int32_t *ext;         // [R][(K1-1)*N + 1]
    int32_t *acc;         // [R][K1][kMir + N] accumulator images
    const cplx *tw1f;     // [H][8][64]: Tables::tw1f (N = 1024) or the two tables of Br2048Args::tw1f2
    const cplx *tw2;      // [8][8]
    Gadget g;
    int32_t n, mu, K1, L, R;
};

template <int NBLK /* N / 64: 16 or 32 */, bool MARGIN = false>
__global__ __launch_bounds__(64, 1) void blind_rotate_kernel_general(BrGenArgs P)
{
    constexpr int N = 64 * NBLK, H = NBLK / 16, kImgN = kMir + N;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *xch = reinterpret_cast<cplx *>(smem);                 // [kXchElems]
    cplx *tw2_lds = xch + kXchElems;                            // [8][8]
    cplx *spec = tw2_lds + 64;                                  // [K1][H][8][64] spectrum accumulators of the step
    const int lane = threadIdx.x;
    const size_t w = blockIdx.x;
    const int K1 = P.K1, L = P.L;
    const int32_t *bara = P.bara + w * (P.n + 1);
    int32_t *acc = P.acc + w * (size_t)K1 * kImgN;
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);

    cplx tw1f[H][8];
#pragma unroll
    for (int h = 0; h < H; h++)
#pragma unroll
        for (int q = 0; q < 8; q++) tw1f[h][q] = P.tw1f[h * 512 + q * 64 + lane];
    tw2_lds[lane] = P.tw2[lane];
    {   // accum = (0, ..., 0, X^{-barb} (mu, ..., mu))     bootstrap.jl:54-56,78
        const int barb = bara[P.n] & (2 * N - 1);
        int32_t v[NBLK];
        for (int c = 0; c < K1; c++) {
#pragma unroll
            for (int m = 0; m < NBLK; m++) {
                const int idx = (lane + 64 * m + barb) & (2 * N - 1);
                v[m] = c + 1 < K1 ? 0 : (idx & N) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
            }
            store_cur<NBLK>(lane, v, acc + c * kImgN);
        }
    }
    auto acc_fence = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    acc_fence();

#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        const int a = bara[i] & (2 * N - 1);
        const cplx *key = P.bk + (size_t)i * ((size_t)L * K1 * K1 * H * kM) + lane;
        for (int j = 0; j < K1 * H * 8; j++) spec[j * 64 + lane] = mk(0.0, 0.0);
        WAVE_LDS_FENCE();
#pragma unroll 1
        for (int c = 0; c < K1; c++) {
            int32_t temp[NBLK];
            rotate_poly<NBLK>(lane, a, acc + c * kImgN, P.g.offset, xormask, temp);
#pragma unroll 1
            for (int p = 0; p < L; p++) {
#pragma unroll
                for (int h = 0; h < H; h++) {
                    cplx x[8];
                    if constexpr (H == 1) {
                        int32_t t16[16];
#pragma unroll
                        for (int m = 0; m < 16; m++) t16[m] = temp[m];
                        load_digits2(t16, p + 1, beta, x);
                    } else {
                        const double sg = h ? -0.70710678118654752440 : 0.70710678118654752440;
                        static_for<0, 8>([&](auto rc) {
                            constexpr int R = decltype(rc)::value;
                            const int32_t lo = digit2(temp[R], p + 1, beta), l2 = digit2(temp[R + 8], p + 1, beta);
                            const int32_t hi = digit2(temp[R + 16], p + 1, beta), h2 = digit2(temp[R + 24], p + 1, beta);
                            x[R] = fwd_in_2048<R>((double)lo, (double)hi, (double)(l2 - h2), (double)(l2 + h2), sg, h != 0);
                        });
                    }
                    fft_fwd_wave(lane, x, tw1f[h], tw2_lds, xch);
                    WAVE_LDS_FENCE();
                    // out[co] += D[p, c] .* BK_i[p, c].a[co]        (tgsw.jl:128)
#pragma unroll 1
                    for (int co = 0; co < K1; co++) {
                        const cplx *kp = key + ((size_t)((p * K1 + c) * K1 + co) * H + h) * kM;
                        cplx *sp = spec + (size_t)(co * H + h) * kM + lane;
#pragma unroll
                        for (int k2 = 0; k2 < 8; k2++) sp[k2 * 64] = cfma(x[k2], kp[k2 * 64], sp[k2 * 64]);
                    }
                    WAVE_LDS_FENCE();
                }
            }
        }
        // every rotated read of this step is done: inverse transforms, rounding, accumulator update (bootstrap.jl:22)
#pragma unroll 1
        for (int co = 0; co < K1; co++) {
            cplx y[H][8];
#pragma unroll
            for (int h = 0; h < H; h++) {
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) y[h][k2] = spec[(size_t)(co * H + h) * kM + k2 * 64 + lane];
                WAVE_LDS_FENCE();
                fft_inv_wave(lane, y[h], tw1f[h], tw2_lds, xch);
                WAVE_LDS_FENCE();
            }
            if constexpr (H == 1) {
                int32_t accr[16];
                load_cur<16>(lane, acc + co * kImgN, accr);
                untwist_add2<MARGIN>(y[0], accr, &worst);
                store_cur<16>(lane, accr, acc + co * kImgN);
            } else {
                finish_2048<MARGIN>(lane, y[0], y[H - 1], acc + co * kImgN, worst);
            }
        }
        acc_fence();
    }

    // tlwe_extract_sample (tlwe.jl:55-59): mask polynomials concatenated in order, b = body[0]
    int32_t *ext = P.ext + w * ((size_t)(K1 - 1) * N + 1);
    for (int c = 0; c + 1 < K1; c++)
#pragma unroll
        for (int m = 0; m < NBLK; m++) {
            const int j = lane + 64 * m;
            const int32_t v = acc[c * kImgN + kMir + j];
            if (j == 0) ext[(size_t)c * N] = v;
            else ext[(size_t)c * N + N - j] = (int32_t)(0u - (uint32_t)v);
        }
    if (lane == 0) ext[(size_t)(K1 - 1) * N] = acc[(K1 - 1) * kImgN + kMir];
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
}
