// kernels_v3.hpp — blind_rotate_kernel_v3: one wave per blind rotation, the dominant kernel (BASELINE configs 2, 3, 4a).
// (bootstrap.jl:19-82, tgsw.jl:99-129)
#pragma once
#include "kernels_common.hpp"

constexpr int kV3SyncEvery = 4;       // CMUX steps between the barriers of a lockstep group (1, 2, 4, 8, 16 measured: 11.62, 11.51, 11.49, 11.54, 11.53 ms)
constexpr int kV3LdsBytes = 2 * kImg * 4 + (kXchElems + 64) * (int)sizeof(cplx);     // per rotation

// v3: one wave per blind rotation at 2 waves/SIMD (<= 256 VGPRs, no AGPR/scratch spills).
//   * pass-A twiddles (with the lane part of the twist folded in) resident in registers, pass-B twiddles
//     in a 1 KB wave-private LDS table, the register part of the twist as compile-time constants:
//     no global loads on the critical path except the key;
//   * the accumulator lives only in LDS (read at rotate time and at the final add), each polynomial with its mirror
//     block so that the rotation's signs and block offsets are scalar (rotate_sub3);
//   * key spectra of the next transform prefetched into registers while the current FFT runs;
//   * wave-private LDS needs only compiler-level ordering, no s_barrier;
//   * RW rotations per workgroup (RW = 1 or 4), one wave each with its own LDS region.  RW = 4: the four waves sit on the
//     four SIMDs of a CU and meet at one s_barrier every kV3SyncEvery steps, so that they stream the SAME 64 KB of key per
//     step at the same time: three of the four reads hit the CU's vector L1 and the exposed key latency (1.8 ms of 11.8
//     when the loads are removed from the single-rotation workgroups) all but disappears (0.2 ms).  The other workgroup of
//     the CU runs free of this one, so each SIMD still holds two waves in different phases.  4096 rotations: 11.5 vs
//     11.8 ms (80-bit), 18.9 vs 19.5 ms (128-bit); below ~2000 rotations (one wave per SIMD, nothing to share the L1 with)
//     the group only costs (1100 rotations: 5.6 vs 5.1 ms; break-even at ~1500): the dispatcher uses RW = 4 from 1536 rotations up.  RW = 2
//     puts the pair on one SIMD pair in the same phase: 13.4 ms; RW = 8: 12.0 ms (profiles/r03/r03r_*, r03t_*);
//   * no branch on bara[i] == 0 (the step then adds exactly zero);
//   * the first transform of a step writes the spectrum accumulators (a product, not a multiply-add): no zeroing.
template <int L, int KPF /* key values prefetched per transform: 16 = whole chunk, 8 = half */, bool TW2REG = false /* pass-B twiddles in registers instead of LDS */,
          bool MARGIN = false /* diagnostics: rounding margin + in-kernel clock (DiagArgs) */, int RW = 1 /* rotations per workgroup */>
__global__ __launch_bounds__(64 * RW, 2) void blind_rotate_kernel_v3(BrArgs P)
{
    constexpr int K1 = 2;
    // KPF == 8: the first half of a transform's key chunk is requested a transform ahead, the first KMID values of the
    // second half inside the transform (between the store and the load of its second transposition, where x[] is dead),
    // the rest after it.  Interleaved A/B on one device, 4096 rotations: l = 2: 12.86 ms against 13.07 with the whole chunk
    // a transform ahead (KPF == 16) and 12.98 with KMID = 0; l = 3: 21.53 against 21.47.  Only <l, 8, tw2reg> is instantiated since round 4.
    constexpr int KMID = (KPF == 8) ? 4 : 0;
    // L = 0: the decomposition length is a run-time value (P.l) — the transform loop is rolled and nothing else depends on it —
    // so ONE instantiation serves every l no shipped parameter set uses at the speed of the tuned ones
    const int Lr = L ? L : P.l;
    const int F = K1 * Lr;
    wave_priority_begin(P.prio_steps);
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
        const int wib = (RW > 1) ? wave_in_block() : 0;
    char *smem = smem_all + (size_t)wib * kV3LdsBytes;
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem);                    // [K1][kImg]
    cplx *xch = reinterpret_cast<cplx *>(smem + K1 * kImg * 4);              // [kXchElems]
    cplx *tw2_lds = xch + kXchElems;                                         // [8][8]
    const int lane = (RW > 1) ? lane_id() : (int)threadIdx.x;
    size_t w = (size_t)blockIdx.x * RW + wib;
    const bool padding = (RW > 1) && w >= (size_t)P.R;                    // recomputes the last rotation, stores nothing
    if (padding) w = P.R - 1;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(Lr, beta);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    tw2_lds[lane] = P.T.tw2[lane];
    cplx tw2r[8];
    if (TW2REG) {
#pragma unroll
        for (int q = 1; q < 8; q++) tw2r[q] = P.T.tw2[q * 8 + (lane & 7)];
    }
    init_zero_poly(lane, acc_lds);
    init_body_poly(lane, bara[P.n] & (2 * kN - 1), P.mu, acc_lds + kImg);
    WAVE_LDS_FENCE();

    double worst = 0.0;
    cplx kbuf[16];
    // chunk f of step: key spectra for transform f = (c, p): 16 values per lane (co-major, k2 minor)
    auto key_ptr = [&](int step, int f) {
        const int c = L ? f / (L ? L : 1) : (f >= Lr), p = L ? f % (L ? L : 1) : f - c * Lr;          // (K1 = 2; the templated l keeps its compile-time division)
        return P.bk + (size_t)step * (Lr * K1 * K1 * kM) + (size_t)(p * K1 + c) * K1 * kM + lane;
    };
    {
        const cplx *kp = key_ptr(0, 0);
#pragma unroll
        for (int j = 0; j < KPF; j++) kbuf[j] = kp[j * 64];
    }

    int a_next = bara[0] & (2 * kN - 1);      // (plain loads here: 0.6 % faster than through the scalar cache in this kernel, measured)
    for (int i = 0; i < P.n; i++) {
        const int a = a_next;
        a_next = bara[i + 1] & (2 * kN - 1);   // bara[n] (= barb) exists: harmless read on the last step
        wave_priority_step(i, P.prio_steps);
        if (RW > 1 && (i % kV3SyncEvery) == 0) __builtin_amdgcn_s_barrier();

        cplx out[K1][8];
        int32_t temp[16];
#pragma unroll 1
        for (int f = 0; f < F; f++) {
            const int c = L ? f / (L ? L : 1) : (f >= Lr), p = L ? f % (L ? L : 1) : f - c * Lr;        // component, digit index (0-based)
            if (p == 0) rotate_poly<16>(lane, a, acc_lds + c * kImg, P.g.offset, xormask, temp);
            cplx x[8];
            load_digits2t(temp, p + 1, beta, x);
            dft8_fwd_tw(x);
            // pass A
#pragma unroll
            for (int q = 0; q < 8; q++) x[q] = cmul(x[q], tw1f[q]);
            x1_store_a(lane, x, xch);
            WAVE_LDS_FENCE();
            x1_load_b(lane, x, xch);
            // pass B (twiddles from the LDS table)
            {
                cplx t2[8];
#pragma unroll
                for (int q = 1; q < 8; q++) t2[q] = TW2REG ? tw2r[q] : tw2_lds[q * 8 + (lane & 7)];
                dft8<false>(x);
#pragma unroll
                for (int q = 1; q < 8; q++) x[q] = cmul(x[q], t2[q]);
            }
            WAVE_LDS_FENCE();
            x2_store(lane, x, xch);
            WAVE_LDS_FENCE();
            cplx k1v[8];
            if (KPF == 8 && KMID > 0) {
                const cplx *kp = key_ptr(i, f);
#pragma unroll
                for (int k2 = 0; k2 < KMID; k2++) k1v[k2] = kp[(8 + k2) * 64];
                WAVE_LDS_FENCE();
            }
            x2_load(lane, x, xch);
            WAVE_LDS_FENCE();
            dft8<false>(x);
            // MAC: out[co] (+)= D[p, c] .* BK_i[p, c].a[co]        (tgsw.jl:128); f is wave-uniform: a scalar branch
            if (KPF == 16) {
                if (f == 0) {
#pragma unroll
                    for (int co = 0; co < K1; co++)
#pragma unroll
                        for (int k2 = 0; k2 < 8; k2++) out[co][k2] = cmul(x[k2], kbuf[co * 8 + k2]);
                } else {
#pragma unroll
                    for (int co = 0; co < K1; co++)
#pragma unroll
                        for (int k2 = 0; k2 < 8; k2++) out[co][k2] = cfma(x[k2], kbuf[co * 8 + k2], out[co][k2]);
                }
            } else {
                const cplx *kp = key_ptr(i, f);
#pragma unroll
                for (int k2 = KMID; k2 < 8; k2++) k1v[k2] = kp[(8 + k2) * 64];
                if (f == 0) {
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[0][k2] = cmul(x[k2], kbuf[k2]);
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[1][k2] = cmul(x[k2], k1v[k2]);
                } else {
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[0][k2] = cfma(x[k2], kbuf[k2], out[0][k2]);
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[1][k2] = cfma(x[k2], k1v[k2], out[1][k2]);
                }
            }
            // prefetch the next transform's key
            {
                const bool last = (f + 1 == F);
                // (unconditional: on the very last transform this re-reads a valid chunk; a conditional
                //  prefetch doubles the register pressure through the phi of old and new values)
                const cplx *kp = last ? key_ptr(i + 1 < P.n ? i + 1 : i, 0) : key_ptr(i, f + 1);
#pragma unroll
                for (int j = 0; j < KPF; j++) kbuf[j] = kp[j * 64];
            }
        }
#pragma unroll
        for (int co = 0; co < K1; co++) {
            dft8<true>(out[co]);
            x2_store(lane, out[co], xch);
            WAVE_LDS_FENCE();
            x2_load(lane, out[co], xch);
            {
                cplx t2[8];
#pragma unroll
                for (int q = 1; q < 8; q++) t2[q] = TW2REG ? tw2r[q] : tw2_lds[q * 8 + (lane & 7)];
#pragma unroll
                for (int q = 1; q < 8; q++) out[co][q] = cmulc(out[co][q], t2[q]);
            }
            dft8<true>(out[co]);
            WAVE_LDS_FENCE();
            x1_store_b(lane, out[co], xch);
            WAVE_LDS_FENCE();
            x1_load_a(lane, out[co], xch);
            WAVE_LDS_FENCE();
#pragma unroll
            for (int q = 0; q < 8; q++) out[co][q] = cmulc(out[co][q], tw1f[q]);
            dft8<true>(out[co]);
            accumulate_poly<MARGIN>(lane, out[co], acc_lds + co * kImg, &worst);
        }
        WAVE_LDS_FENCE();
    }

    if (padding) return;
    int32_t *ext = P.ext + w * (kN + 1);
    extract_mask_poly(lane, acc_lds, ext);
    if (lane == 0) ext[kN] = acc_lds[kImg + kMir];
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
}
