// kernels_k2.hpp — tlwe_mask_size k = 2 (api.jl:30,55): blind_rotate_kernel_k2 (one wave per rotation) and blind_rotate_kernel_k2w3 (three).
#pragma once
#include "kernels_common.hpp"
#include "kernels_v3.hpp"      // kV3SyncEvery: the lockstep groups of blind_rotate_kernel_k2 follow v3's

// ---- blind rotation for tlwe_mask_size k = 2 (api.jl:30,55 keyword) ---------------------------------
// (This kernel keeps the accumulator polynomials WITHOUT mirror blocks and rotates with per-lane signs, rotate_sub2: the three
// mirrors would take its LDS from 22.5 to 23.3 KB per wave, i.e. from seven to six waves per CU — measured 29.0 vs 28.4 ms per 4096 rotations.)
// Same algorithm as blind_rotate_kernel_v3 with a 3-polynomial accumulator: 3*L forward transforms and
// 3 inverse transforms per step, out[co] += D[p, c] .* BK_i[p, c].a[co] for c, co in 0..2 (tgsw.jl:125-129).
constexpr int kK2LdsBytes = 3 * kN * 4 + (kXchElems + 64) * (int)sizeof(cplx);      // per rotation
template <int L, bool MARGIN = false, int RW = 1 /* rotations per workgroup, in lockstep (as blind_rotate_kernel_v3) */>
__global__ __launch_bounds__(64 * RW, 2) void blind_rotate_kernel_k2(BrArgs P)
{
    constexpr int K1 = 3;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int wib = (RW > 1) ? wave_in_block() : 0;
    char *smem = smem_all + (size_t)wib * kK2LdsBytes;
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem);                    // [K1][N]
    cplx *xch = reinterpret_cast<cplx *>(smem + K1 * kN * 4);
    cplx *tw2_lds = xch + kXchElems;
    const int lane = (RW > 1) ? lane_id() : (int)threadIdx.x;
    // RW > 1: the batch is dealt out in whole rounds of one workgroup per CU, every workgroup with grp_q or grp_q + 1 (<= RW)
    // rotations so that the rounds are equally full; the waves beyond a workgroup's count only keep the barriers company
    size_t w = blockIdx.x;
    if (RW > 1) {
        const int g = (int)blockIdx.x;
        const int cnt = g < P.grp_big ? P.grp_q + 1 : P.grp_q;
        const size_t base = g < P.grp_big ? (size_t)g * (P.grp_q + 1) : (size_t)P.grp_big * (P.grp_q + 1) + (size_t)(g - P.grp_big) * P.grp_q;
        if (wib >= cnt) {
            for (int i = 0; i < P.n; i += kV3SyncEvery) __builtin_amdgcn_s_barrier();
            return;
        }
        w = base + wib;
    }
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    tw2_lds[lane] = P.T.tw2[lane];
    {
        const int barb = bara[P.n] & (2 * kN - 1);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int idx = (lane + 64 * m + barb) & (2 * kN - 1);
            acc_lds[lane + 64 * m] = 0;
            acc_lds[kN + lane + 64 * m] = 0;
            acc_lds[2 * kN + lane + 64 * m] = (idx & kN) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
    }
    WAVE_LDS_FENCE();

    // (no wave_priority_* here: 22 KB of LDS per wave put 7 waves on a CU, so one SIMD has a single wave; measured 3 % slower with it)
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        // (a plain load, not load_uniform_i32: with the exponent arriving through the scalar cache every wave of this kernel
        //  takes the same time to the microsecond, and 4096 rotations on 7 x 256 wave slots then run as three strict rounds —
        //  37.0 ms against 28.6 ms with the natural spread of the waves' progress; measured, profiles/r03/r03p_k2_exponent_load.txt)
        const int a = bara[i] & (2 * kN - 1);
        if (RW > 1 && (i % kV3SyncEvery) == 0) __builtin_amdgcn_s_barrier();
        const cplx *key = P.bk + (size_t)i * (L * K1 * K1 * kM) + lane;
        // (zeroed, then accumulated.  Writing the first transform's products instead — a peeled first iteration of both loops, as
        //  blind_rotate_kernel_w2 does — triples the loop body: 30.0 against 26.3 ms per 4096 rotations, measured.)
        cplx out[K1][8];
#pragma unroll
        for (int d = 0; d < K1; d++)
#pragma unroll
            for (int q = 0; q < 8; q++) out[d][q] = mk(0.0, 0.0);
#pragma unroll 1
        for (int c = 0; c < K1; c++) {
            int32_t temp[16];
            {
                int32_t cur[16];
#pragma unroll
                for (int m = 0; m < 16; m++) cur[m] = acc_lds[c * kN + lane + 64 * m];
                int a_here = a;
                asm volatile("" : "+v"(a_here));
                rotate_sub2(lane, a_here, acc_lds + c * kN, cur, P.g.offset, xormask, temp);
            }
#pragma unroll 1
            for (int p = 0; p < L; p++) {
                const cplx *kp = key + (size_t)(p * K1 + c) * K1 * kM;
                cplx kfirst[8];                               // co = 0 requested before the FFT
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) kfirst[k2] = kp[k2 * 64];
                cplx x[8];
                load_digits2t(temp, p + 1, beta, x);
                fft_fwd_wave<true>(lane, x, tw1f, tw2_lds, xch);
#pragma unroll
                for (int co = 0; co < K1; co++) {
                    cplx kv[8];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) kv[k2] = co == 0 ? kfirst[k2] : kp[(co * 8 + k2) * 64];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[co][k2] = cfma(x[k2], kv[k2], out[co][k2]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d < K1; d++) {
            fft_inv_wave(lane, out[d], tw1f, tw2_lds, xch);
            int32_t accr[16];
#pragma unroll
            for (int m = 0; m < 16; m++) accr[m] = acc_lds[d * kN + lane + 64 * m];
            untwist_add2<MARGIN>(out[d], accr, &worst);
            for (int m = 0; m < 16; m++) acc_lds[d * kN + lane + 64 * m] = accr[m];
        }
        WAVE_LDS_FENCE();
    }
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
    // tlwe_extract_sample (tlwe.jl:55-59): mask polynomials concatenated in order, b = body[0]
    int32_t *ext = P.ext + w * (2 * kN + 1);
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int jj = lane + 64 * m;
            const int32_t v = acc_lds[c * kN + jj];
            if (jj == 0) ext[c * kN] = v;
            else ext[c * kN + kN - jj] = (int32_t)(0u - (uint32_t)v);
        }
    if (lane == 0) ext[2 * kN] = acc_lds[2 * kN];
}

// ---- k = 2, small batches and the last round of a large one: THREE waves per blind rotation (round 5) ------------------
// blind_rotate_kernel_k2 runs a rotation on one wave: 3 L forward and 3 inverse transforms per CMUX step back to back, and a
// round of up to four rotations per CU — one wave per SIMD — costs the same 6.8 - 7.3 ms however few rotations it holds
// (kK2RoundCost, engine_dispatch.hip): a batch of 4096 = 16 per CU pays 7.3 ms for its last four.  Here wave c owns accumulator
// polynomial c (blind_rotate_kernel_w2's structure with three polynomials): it rotates and decomposes only its own polynomial,
// runs its L forward transforms, multiplies each spectrum into partial sums of all three output components, keeps its own
// and hands the other two over — the one for wave c + 1 through its transposition buffer, which changes hands (after the
// barrier wave c + 1 reads it and keeps it for its inverse transform and the next step's forward transforms: buffer of wave
// c in step i = (c - i) mod 3), the one for wave c + 2 through a hand-off slot of its own — then adds the two partial sums
// it receives, inverse-transforms its component and updates its polynomial: L + 1 transforms per wave and step instead of
// 3 L + 3.  Two barriers per step (the second keeps a slot's reader ahead of its next writer).  64.4 KB of LDS per rotation:
// two rotations per CU (six waves), so this kernel takes batches of up to two rotations per CU and the last round of a
// larger one when that is what is left (k2_partition).
constexpr int kK2W3LdsBytes = 3 * kImg * 4 + (3 * kXchElems + 3 * kM + 64) * (int)sizeof(cplx);      // per rotation
template <int L, bool MARGIN = false>
__global__ __launch_bounds__(192, 2) void blind_rotate_kernel_k2w3(BrArgs P)
{
    constexpr int K1 = 3;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_all = reinterpret_cast<int32_t *>(smem);                        // [K1][kImg]
    cplx *xch_all = reinterpret_cast<cplx *>(smem + K1 * kImg * 4);              // [3][kXchElems]: handed round every step
    cplx *slot_all = xch_all + 3 * kXchElems;                                    // [3][kM]: wave c's partial sum for wave c + 2
    cplx *tw2_lds = slot_all + 3 * kM;                                           // [8][8]
    const int wv = wave_in_block();                                              // wave = owned polynomial (scalar)
    const int lane = (int)threadIdx.x & 63;
    const int wn1 = wv == 2 ? 0 : wv + 1, wn2 = wv == 0 ? 2 : wv - 1;            // (wv + 1) mod 3, (wv + 2) mod 3
    int32_t *acc_lds = acc_all + wv * kImg;
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    if (threadIdx.x < 64) tw2_lds[threadIdx.x] = P.T.tw2[threadIdx.x];
    if (wv == 2) init_body_poly(lane, bara[P.n] & (2 * kN - 1), P.mu, acc_lds);
    else init_zero_poly(lane, acc_lds);
    __syncthreads();

    int a_next = load_uniform_i32(bara) & (2 * kN - 1);
    int b = wv;                                                                  // this wave's buffer: (wv - i) mod 3
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        const int a = a_next;
        a_next = load_uniform_i32(bara + i + 1) & (2 * kN - 1);   // bara[n] (= barb) exists: harmless read on the last step
        // key polys of transform (p, c = wv): [i][p][c][co][8][64]
        const cplx *key = P.bk + (size_t)i * (L * K1 * K1 * kM) + (size_t)wv * K1 * kM + lane;
        const int bprev = b == 0 ? 2 : b - 1;                                    // buffer of wave wv - 1 in this step
        cplx *xch = xch_all + b * kXchElems, *xch_next = xch_all + bprev * kXchElems;
        cplx own[8], o1[8], o2[8];
        int32_t temp[16];
        rotate_poly<16>(lane, a, acc_lds, P.g.offset, xormask, temp);
        auto digit = [&](int p, auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            cplx x[8];
            load_digits2t(temp, p + 1, beta, x);
            const cplx *kp = key + (size_t)p * K1 * K1 * kM;
            cplx kv[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) kv[k2] = kp[(size_t)wv * kM + k2 * 64];             // co = wv (issued before the FFT)
            fft_fwd_wave<true>(lane, x, tw1f, tw2_lds, xch);
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) own[k2] = FIRST ? cmul(x[k2], kv[k2]) : cfma(x[k2], kv[k2], own[k2]);
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) kv[k2] = kp[(size_t)wn1 * kM + k2 * 64];            // co = wv + 1
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) o1[k2] = FIRST ? cmul(x[k2], kv[k2]) : cfma(x[k2], kv[k2], o1[k2]);
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) kv[k2] = kp[(size_t)wn2 * kM + k2 * 64];            // co = wv + 2
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) o2[k2] = FIRST ? cmul(x[k2], kv[k2]) : cfma(x[k2], kv[k2], o2[k2]);
        };
        digit(0, std::true_type{});
#pragma unroll 1
        for (int p = 1; p < L; p++) digit(p, std::false_type{});
        // hand the other two components' partial sums over
        WAVE_LDS_FENCE();
        cplx *slot = slot_all + wv * kM;
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) { xch[k2 * 64 + lane] = o1[k2]; slot[k2 * 64 + lane] = o2[k2]; }
        __syncthreads();
        const cplx *from2 = slot_all + wn1 * kM;                                 // wave wv + 1's partial sum for wave (wv + 1) + 2 = wv
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) own[k2] = cadd(own[k2], cadd(xch_next[k2 * 64 + lane], from2[k2 * 64 + lane]));
        WAVE_LDS_FENCE();
        __syncthreads();                 // every slot has been read: its owner may write it again in the next step
        fft_inv_wave(lane, own, tw1f, tw2_lds, xch_next);
        accumulate_poly<MARGIN>(lane, own, acc_lds, &worst);
        WAVE_LDS_FENCE();
        b = bprev;
    }
    __syncthreads();
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
    // tlwe_extract_sample (tlwe.jl:55-59): mask polynomials concatenated in order, b = body[0]
    int32_t *ext = P.ext + w * (2 * kN + 1);
    if (wv < 2) extract_mask_poly(lane, acc_lds, ext + (size_t)wv * kN);
    else if (lane == 0) ext[2 * kN] = acc_lds[kMir];
}
