// kernels_multikey.hpp — multi-key blind rotation: mk_w2 (2 parties, config 5), mk_general (any party count / length), mk_g2 (the
// shipped 4- and 8-party sets).  (mk_internals.jl:348-391,464-495)
#pragma once
#include "kernels_common.hpp"

// ---- multi-key blind rotation (2 parties) ----------------------------------------------------------
// mk_internals.jl:464-495 (mk_mux_rotate, mk_blind_rotate, extract) with mk_tgsw_extern_mul (:348-391).
// Accumulator = P mask polynomials + body (P = 2): 3 polynomials in LDS.  Per step (party i, bit j):
// 3*L forward transforms, MAC against the expanded key polys x, y, c0, c1 of (i, j), 3 inverse
// transforms.  The reference inverse-transforms every product separately and sums in Int32
// (:359-366); summing in the spectrum domain first gives the same words (both are the exact product
// mod 2^32; rounding margin checked by the oracle test).
struct MkBrArgs {
    DiagArgs diag;
    int32_t R;            // rotations in the batch (workgroups of mk_blind_rotate_kernel_w2 hold several: the last one may be padded)
    const int32_t *bara;  // [R][P*n+1]
    const cplx *bk;       // [P][n][2*L*P + 2*L][8][64] spectra (engine order, scaled 1/M)
    int32_t *ext;         // [R][P*N+1]
    Tables T;
    Gadget g;
    int32_t n;
    int32_t mu;
    int32_t prio_steps;   // of the P * n steps of a rotation; see wave_priority_begin
};

// ---- multi-key blind rotation, 2 parties, TWO waves per rotation ------------------------------------------------
// BASELINE config 5 is 1024 rotations: with one wave per rotation that is ONE wave per SIMD (a lone wave issues FP64 at
// about half the SIMD's rate) running 12 forward and 3 inverse transforms per step back to back.  Here the two waves of
// a workgroup split the 3 L forward transforms of a step evenly (wave 0: every digit of the party's mask and half the
// digits of the other mask; wave 1: every digit of the body and the other half) and multiply them into their own
// partial sums of the three new polynomials (mk_internals.jl:371-385); the partial sums are handed over through LDS (wave 1 gives the two mask partials to
// wave 0, wave 0 the body partial to wave 1), each owner adds what it receives, inverse-transforms and updates its
// polynomials.  Two synchronisations of the pair per step (hand-off written / accumulator updated: pair_signal, kernels_common.hpp).  All 1024 rotations are resident at two waves per SIMD (39.4 KB of LDS per workgroup: the hand-off
// reuses the transposition buffers).  Same words as the any-party kernel (round 3's one-wave 2-party kernel is gone).  L must be even.
template <int L, int PARTY, int WV, bool MARGIN, int TAN, bool PAIR>
__device__ __forceinline__ void mk2_party_steps(int lane_in, const MkBrArgs &P, const int32_t *bara, int32_t *acc_lds,
                                                cplx *xch_own, cplx *xch_oth, cplx *extra, const cplx *tw2_lds, const cplx (&tw1f)[8],
                                                int32_t xormask, double &worst, const Tan16 &tk, int *pair_flags)
{
    constexpr int NP = 2;
    constexpr int PER = 2 * L * NP + 2 * L;       // key polys per (party, bit): x[L][NP] | y[L][NP] | c0[L] | c1[L]
    constexpr int OTHER = 1 - PARTY;
    constexpr int MKPN = 2;
    const int beta = P.g.log2_base;
    int a_next = load_uniform_i32(bara + PARTY * P.n) & (2 * kN - 1);
    STAMP_DECL;
#pragma unroll 1
    for (int j = 0; j < P.n; j++) {
        wave_priority_step(PARTY * P.n + j, P.prio_steps);
        const int g = PARTY * P.n + j;            // step of the rotation: its two hand-offs are numbered 2 g + 1 and 2 g + 2 (pair_signal)
        if (PAIR && (g & (kPairSyncEvery - 1)) == 0) __syncthreads();
        const int a = a_next;
        a_next = load_uniform_i32(bara + PARTY * P.n + j + 1) & (2 * kN - 1);      // the row ends with barb: the read past the last bit is in range
        // (the lane is made opaque once per step: per-lane addresses — the 64-bit key pointer, the LDS transposition and
        //  accumulator offsets — are then rebuilt from scalar bases here instead of living, and being spilled, across the
        //  whole loop)
        (void)lane_in;
        const int lane = lane_id_fresh();
        const cplx *key = P.bk + ((size_t)PARTY * P.n + j) * PER * kM + lane;
        cplx out[NP + 1][8];                      // partial sums of the new a_0, a_1, b over this wave's transforms
#pragma unroll
        for (int d = 0; d <= NP; d++)
#pragma unroll
            for (int q = 0; q < 8; q++) out[d][q] = mk(0.0, 0.0);
        // Work split (12 forward transforms per step): wave 0 takes all L digits of the party's mask and digits
        // [0, L/2) of the other mask, wave 1 all L digits of the body and digits [L/2, L) of the other mask: each wave
        // rotates and decomposes two source polynomials instead of three.
        static_for<0, 2>([&](auto job_c) {
            constexpr int job = decltype(job_c)::value;
            constexpr int s = job == 0 ? (WV == 0 ? PARTY : NP) : OTHER;      // source polynomial
            constexpr int p_begin = job == 0 ? 0 : WV * (L / 2), p_end = job == 0 ? L : (WV + 1) * (L / 2);
            int32_t temp[16];
            rotate_poly<16>(lane, a, acc_lds + s * kImg, P.g.offset, xormask, temp);
            STAMP(0);
#pragma unroll 1
            for (int p = p_begin; p < p_end; p++) {
                const cplx *k_party, *k_body;
                if (s < NP) {
                    k_party = key + (size_t)(L * NP + p * NP + s) * kM;         // y[p, s]      -> a'_party
                    k_body = key + (size_t)(p * NP + s) * kM;                   // x[p, s]      -> b'
                } else {
                    k_party = key + (size_t)(2 * L * NP + L + p) * kM;          // c1[p]        -> a'_party
                    k_body = key + (size_t)(2 * L * NP + p) * kM;               // c0[p]        -> b'
                }
                const cplx *k_other = key + (size_t)(L * NP + p * NP + PARTY) * kM;           // y[p, party] -> a'_other (s == OTHER only)
                cplx kpa[8];                      // requested before the FFT (a second poly in flight spills)
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) kpa[k2] = k_party[k2 * 64];
                cplx x[8];
                if constexpr (TAN) { load_digits2t_o(temp, p + 1, beta, x, tk); dft8_fwd_tw_o(x, tk); }
                else load_digits2(temp, p + 1, beta, x);
                cplx kbo[8];
                // (the first values of the second poly are requested inside the transform, where x[] is dead: see blind_rotate_kernel_n2048x)
                fft_fwd_wave_mid<false, (TAN != 0)>(lane, x, tw1f, tw2_lds, xch_own, [&]() {
#pragma unroll
                    for (int k2 = 0; k2 < MKPN; k2++) kbo[k2] = k_body[k2 * 64];
                });
                STAMP(1);
                // The second key polynomial arrives in two halves: the rest of its first half now, its second half (and the third
                // polynomial, into kpa's registers) only after the first product has consumed kpa — with all of it requested
                // at once the step held 270 values live and spilled (60 B / lane of scratch in round 2).
#pragma unroll
                for (int k2 = MKPN; k2 < 4; k2++) kbo[k2] = k_body[k2 * 64];
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) out[PARTY][k2] = cfma(x[k2], kpa[k2], out[PARTY][k2]);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k2 = 4; k2 < 8; k2++) kbo[k2] = k_body[k2 * 64];
                if (s == OTHER) {
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) kpa[k2] = k_other[k2 * 64];
                }
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) out[NP][k2] = cfma(x[k2], kbo[k2], out[NP][k2]);
                if (s == OTHER) {
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[OTHER][k2] = cfma(x[k2], kpa[k2], out[OTHER][k2]);
                }
                STAMP(2);
            }
        });
        // hand-off: each wave writes what the other one owns into its OWN transposition buffer (+ the extra slot), so
        // nothing of the other wave's is touched before the barrier
        if (WV == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) xch_own[k2 * 64 + lane] = out[NP][k2];
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) { xch_own[k2 * 64 + lane] = out[0][k2]; extra[k2 * 64 + lane] = out[1][k2]; }
        }
        STAMP(3);
        {
            constexpr int NG = WV == 0 ? 2 : 1;       // wave 0 receives the two mask partials, wave 1 the body partial
            cplx got[NG][8];
            auto take = [&]() {
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) {
                    got[0][k2] = xch_oth[k2 * 64 + lane];
                    if (WV == 0) got[NG - 1][k2] = extra[k2 * 64 + lane];
                }
            };
            if (PAIR) { pair_signal(pair_flags + WV, 2 * g + 1); pair_wait_take(pair_flags + (1 - WV), 2 * g + 1, take); }
            else { __syncthreads(); take(); }
            STAMP(4);
            if (WV == 0) {
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) { out[0][k2] = cadd(out[0][k2], got[0][k2]); out[1][k2] = cadd(out[1][k2], got[NG - 1][k2]); }
            } else {
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) out[NP][k2] = cadd(out[NP][k2], got[0][k2]);
            }
        }
        STAMP(5);
        // No barrier here: the inverse transforms run in the OTHER wave's transposition buffer, the one this wave has just
        // read (a wave's LDS operations execute in order) and that its owner does not touch again before the barrier that
        // ends the step; this wave's own buffer may still be being read by the other wave.
        WAVE_LDS_FENCE();
        STAMP(6);
        auto finish = [&](cplx (&o)[8], int d) {
            fft_inv_wave(lane, o, tw1f, tw2_lds, xch_oth);
            if constexpr (TAN == 2) {
                int32_t accr[16];
                load_cur<16>(lane, acc_lds + d * kImg, accr);
                untwist_add2_o<MARGIN>(o, accr, &worst, tk);
                store_cur<16>(lane, accr, acc_lds + d * kImg);
            } else {
                accumulate_poly<MARGIN, false>(lane, o, acc_lds + d * kImg, &worst);
            }
        };
        if (WV == 0) { finish(out[0], 0); finish(out[1], 1); }
        else finish(out[NP], NP);
        STAMP(7);
        // the updated accumulator is visible to both waves' rotations of the next step, and the buffer this wave transformed in is its owner's again
        if (PAIR) { pair_signal(pair_flags + WV, 2 * g + 2); pair_wait(pair_flags + (1 - WV), 2 * g + 2); } else __syncthreads();
        STAMP(8);
    }
    if (PARTY == 1) STAMP_FLUSH(P.diag, WV);
}

// RW rotations per workgroup advance in lockstep (the barriers are workgroup-wide): rotations that read the same key
// values at the same time share one trip beyond L2 (the 2-party key is 197 MB as spectra).
template <int L, bool MARGIN = false, int RW = 2>
__global__ __launch_bounds__(128 * RW, 2) void mk_blind_rotate_kernel_w2(MkBrArgs P)
{
    // Round 6: the register part of the twist in tan form, forward and inverse, with OPAQUE constants (load_tan16): 16.54 -> 16.28 ms per
    // 1024 gates on one device (forward only: 16.36).  Round 5 tried the same with compile-time constants and lost 23 % to scalar spills.
    constexpr int TAN = 2;
    static_assert(L % 2 == 0, "the two waves split the digits evenly");
    constexpr int NP = 2;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = lane_id(), wib = wave_in_block(), wv = wib & 1;
    const int rot = wib >> 1;                                                    // rotation within the workgroup
    constexpr size_t kRotBytes = (NP + 1) * kImg * 4 + (2 * kXchElems + kM) * sizeof(cplx);
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem + rot * kRotBytes);      // [NP+1][kImg]
    cplx *xch_all = reinterpret_cast<cplx *>(smem + rot * kRotBytes + (NP + 1) * kImg * 4);   // [2 waves][kXchElems]
    cplx *extra = xch_all + 2 * kXchElems;                                       // [512] second hand-off slot of wave 1
    cplx *tw2_lds = reinterpret_cast<cplx *>(smem + RW * kRotBytes);             // [8][8]
    int *pair_flags = reinterpret_cast<int *>(smem + RW * kRotBytes + 64 * sizeof(cplx)) + rot * 2;      // [RW][2]: pair_signal
    constexpr bool PAIR = RW > 1;      // (one rotation per workgroup: the pair is the workgroup, barriers)
    cplx *xch_own = xch_all + wv * kXchElems, *xch_oth = xch_all + (1 - wv) * kXchElems;
    const size_t w_raw = (size_t)blockIdx.x * RW + rot;
    const bool live = w_raw < (size_t)P.R;                                       // a padding rotation repeats the last one, stores nothing
    const size_t w = live ? w_raw : (size_t)P.R - 1;
    const int32_t *bara = P.bara + w * (NP * P.n + 1);
    const int32_t xormask = gadget_xor_mask(L, P.g.log2_base);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    if (wib == 0) tw2_lds[lane] = P.T.tw2[lane];
    if (lane == 0) pair_flags[wv] = 0;
    // acc = (0, ..., 0, X^{-barb} * mu)       mk_internals.jl:491-492, 72-79
    if (wv == 0) { init_zero_poly(lane, acc_lds); init_zero_poly(lane, acc_lds + kImg); }
    else init_body_poly(lane, bara[NP * P.n] & (2 * kN - 1), P.mu, acc_lds + 2 * kImg);
    __syncthreads();
    wave_priority_begin(P.prio_steps);
    Tan16 tk;
    if constexpr (TAN != 0) tk = load_tan16<(TAN == 2)>();
    // party-major double loop (mk_internals.jl:475-476)
    if (wv == 0) {
        mk2_party_steps<L, 0, 0, MARGIN, TAN, PAIR>(lane, P, bara, acc_lds, xch_own, xch_oth, extra, tw2_lds, tw1f, xormask, worst, tk, pair_flags);
        mk2_party_steps<L, 1, 0, MARGIN, TAN, PAIR>(lane, P, bara, acc_lds, xch_own, xch_oth, extra, tw2_lds, tw1f, xormask, worst, tk, pair_flags);
    } else {
        mk2_party_steps<L, 0, 1, MARGIN, TAN, PAIR>(lane, P, bara, acc_lds, xch_own, xch_oth, extra, tw2_lds, tw1f, xormask, worst, tk, pair_flags);
        mk2_party_steps<L, 1, 1, MARGIN, TAN, PAIR>(lane, P, bara, acc_lds, xch_own, xch_oth, extra, tw2_lds, tw1f, xormask, worst, tk, pair_flags);
    }
    if (!live) return;
    const int lane_e = lane_id_fresh();
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0, wv == 0 && lane_e == 0);
    // mk_tlwe_extract_sample (mk_internals.jl:88-95): one extracted mask column per party, b = body[0]
    int32_t *ext = P.ext + w * (NP * kN + 1);
    extract_mask_poly(lane_e, acc_lds + wv * kImg, ext + wv * kN);                // wave c extracts mask column c
    if (wv == 0 && lane_e == 0) ext[NP * kN] = acc_lds[NP * kImg + kMir];
}

// ---- multi-key blind rotation, any number of parties (2..8) and any decomposition length (<= 8) ------------
// The multi-key blind rotation (header above MkBrArgs) with run-time P and L.  Only three spectrum accumulators are ever
// live whatever P is: in step (party i, bit j) the new mask a'_s of a non-party s receives products of its OWN
// digits only (mk_internals.jl:377-378), so it is inverse-transformed and written back right after source s's
// L transforms (nobody else reads acc[s] in this step); a'_party and b' accumulate over all sources
// (:371-376, :382-385).
struct MkGenArgs {
    DiagArgs diag;
    int32_t R;            // rotations in the batch (a workgroup holds RW of them: the last one may be padded)
    const int32_t *bara;  // [R][P*n+1]
    const cplx *bk;       // [P][n][2*L*P + 2*L][8][64]
    int32_t *ext;         // [R][P*N+1]
    Tables T;
    Gadget g;
    int32_t n, mu, parties, L;
    int32_t *acc;         // ACCG only: [rotations rounded up to the workgroup size][P+1][N] accumulators in global memory
    int32_t prio_steps;   // mk_blind_rotate_kernel_g2: of the P * n steps of a rotation; see wave_priority_begin
};

// RW rotations (one wave each) per workgroup, kept in lockstep by one barrier per CMUX step: the 4- and 8-party keys are
// 0.8 and 4.7 GB as spectra (1.15 MB per step at 8 parties), far beyond L2 and the Infinity Cache, and rotations that
// read the same key values at the same time share one trip to HBM.  Nothing is exchanged between the waves.
// ACCG: the accumulator (P + 1 polynomials, 36 KB at 8 parties) lives in global memory instead of LDS, so that LDS (10 KB
// per wave) no longer limits a CU to 3 rotations at 8 parties / 5 at 4: the accumulator traffic (two reads and one write
// of every polynomial per step, L2-resident) is a tenth of the step's key traffic.  A wave reads back only what it wrote
// itself; the workgroup-scope fence at the end of a step orders those stores before the next step's loads.
template <bool MARGIN = false, int RW = 1, bool ACCG = false>
__global__ __launch_bounds__(64 * RW, 1) void mk_blind_rotate_kernel_general(MkGenArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NP = P.parties, L = P.L;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    const int lane = threadIdx.x & 63, rot = wave_in_block();
    const size_t w_raw = (size_t)blockIdx.x * RW + rot;
    const size_t rot_bytes = (ACCG ? 0 : (size_t)(NP + 1) * kImg * 4) + (kXchElems + 64) * sizeof(cplx);
    int32_t *acc_lds = ACCG ? P.acc + w_raw * (size_t)(NP + 1) * kImg                // [NP+1][kImg] (the name stays: LDS in the default build)
                            : reinterpret_cast<int32_t *>(smem + rot * rot_bytes);
    cplx *xch = reinterpret_cast<cplx *>(smem + rot * rot_bytes + (ACCG ? 0 : (size_t)(NP + 1) * kImg * 4));
    cplx *tw2_lds = xch + kXchElems;
    const bool live = w_raw < (size_t)P.R;                                   // a padding rotation repeats the last one, stores nothing
    const size_t w = live ? w_raw : (size_t)P.R - 1;
    const int32_t *bara = P.bara + w * ((size_t)NP * P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);
    const int per = 2 * L * NP + 2 * L;

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    tw2_lds[lane] = P.T.tw2[lane];
    for (int s = 0; s < NP; s++) init_zero_poly(lane, acc_lds + s * kImg);
    init_body_poly(lane, bara[(size_t)NP * P.n] & (2 * kN - 1), P.mu, acc_lds + NP * kImg);
    auto acc_fence = [&]() {
        if (ACCG) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        } else {
            WAVE_LDS_FENCE();
        }
    };
    acc_fence();

    auto finish = [&](cplx (&o)[8], int d) {      // inverse transform, round, add into accumulator polynomial d
        fft_inv_wave(lane, o, tw1f, tw2_lds, xch);
        accumulate_poly<MARGIN>(lane, o, acc_lds + d * kImg, &worst);
    };

#pragma unroll 1
    for (int party = 0; party < NP; party++) {                               // mk_internals.jl:475
#pragma unroll 1
        for (int j = 0; j < P.n; j++) {                                      // :476
            const int a = bara[(size_t)party * P.n + j] & (2 * kN - 1);
            const cplx *key = P.bk + ((size_t)party * P.n + j) * per * kM + lane;
            cplx o_party[8], o_body[8];
#pragma unroll
            for (int q = 0; q < 8; q++) { o_party[q] = mk(0.0, 0.0); o_body[q] = mk(0.0, 0.0); }
#pragma unroll 1
            for (int s = 0; s <= NP; s++) {
                const bool is_body = (s == NP), has_self = (!is_body && s != party);
                cplx o_self[8];
#pragma unroll
                for (int q = 0; q < 8; q++) o_self[q] = mk(0.0, 0.0);
                int32_t temp[16];
                rotate_poly<16>(lane, a, acc_lds + s * kImg, P.g.offset, xormask, temp);
#pragma unroll 1
                for (int p = 0; p < L; p++) {
                    const cplx *k_party = key + (size_t)(is_body ? 2 * L * NP + L + p : L * NP + p * NP + s) * kM;   // c1[p] | y[p, s]
                    const cplx *k_body = key + (size_t)(is_body ? 2 * L * NP + p : p * NP + s) * kM;                 // c0[p] | x[p, s]
                    // One wave per SIMD (a whole SIMD's registers, the surplus used as spill space): the two key polys every
                    // source needs are requested before the transform, the third after it.  (At two waves per SIMD with 256
                    // registers the kernel spills to scratch inside the transform loop: 3x slower, measured.)
                    cplx kpa[8], kbo[8];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) { kpa[k2] = k_party[k2 * 64]; kbo[k2] = k_body[k2 * 64]; }
                    cplx x[8];
                    load_digits2(temp, p + 1, beta, x);
                    fft_fwd_wave(lane, x, tw1f, tw2_lds, xch);
                    cplx kv[8];
                    if (has_self) {
                        const cplx *k_self = key + (size_t)(L * NP + p * NP + party) * kM;                           // y[p, party]
#pragma unroll
                        for (int k2 = 0; k2 < 8; k2++) kv[k2] = k_self[k2 * 64];
                    }
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) o_party[k2] = cfma(x[k2], kpa[k2], o_party[k2]);
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) o_body[k2] = cfma(x[k2], kbo[k2], o_body[k2]);
                    if (has_self) {
#pragma unroll
                        for (int k2 = 0; k2 < 8; k2++) o_self[k2] = cfma(x[k2], kv[k2], o_self[k2]);
                    }
                }
                if (has_self) finish(o_self, s);     // a'_s complete: only source s feeds it, only source s read acc[s]
            }
            finish(o_party, party);
            finish(o_body, NP);
            acc_fence();
            if (RW > 1) __syncthreads();     // lockstep only: the workgroup's rotations share their key fetches
        }
    }
    if (!live) return;

    int32_t *ext = P.ext + w * ((size_t)NP * kN + 1);
    for (int c = 0; c < NP; c++) extract_mask_poly(lane, acc_lds + c * kImg, ext + (size_t)c * kN);
    if (lane == 0) ext[(size_t)NP * kN] = acc_lds[NP * kImg + kMir];
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0, lane == 0);
}

// ---- multi-key blind rotation for the shipped 4- and 8-party sets: compile-time (parties, l), TWO waves per rotation ----
// mktfhe_parameters_4party / _8party (mk_api.jl:16-34) at 1024 rotations are one wave per SIMD with one wave per rotation,
// and the any-party kernel above additionally needs a whole SIMD's registers.  Here a rotation is two waves that split the
// (P + 1) l forward transforms of a step by SOURCE polynomial, so that the new mask a'_s of a non-party source s — fed by
// its own digits only (mk_internals.jl:377-378) — is finished entirely inside the wave that owns s and nothing of it is
// exchanged:
//     wave 0: the first n0 = ceil((P-1)/2) non-party sources, digits [0, d0) of the body;
//     wave 1: the other n1 non-party sources, the party's own mask, digits [d0, l) of the body
// (d0 balances the transform counts of the two waves: 12 + 3 vs 13 + 2 at 4 parties, 36 + 5 vs 36 + 4 at 8).  Each wave
// keeps partial sums of a'_party and b' over its sources; at the end of the step wave 1 hands its a'_party partial to
// wave 0 and wave 0 its b' partial to wave 1 (each through its own transposition buffer; the inverse transforms then run
// in the other wave's buffer as in mk_blind_rotate_kernel_w2: two barriers per step).  The accumulator (P + 1 polynomial
// images) lives in global memory, L2-resident, as in the any-party kernel's ACCG variant: LDS holds the two transposition
// buffers only (18.4 KB per rotation), so all 1024 rotations are resident at two waves per SIMD; a wave reads a polynomial
// another wave wrote only across the end-of-step barrier + workgroup-scope fence.  RW rotations per workgroup advance in
// lockstep and share their key fetches (the 8-party key is 4.7 GB as spectra).
// Everything that depends on (party, wave, source) is a compile-time constant — one copy of the step per (party, wave), the
// sources unrolled inside it — exactly as in the 2-party kernel: a first version with run-time source lists and ONE copy of
// the step needed its pass-A twiddles and the decomposed source in LDS to fit 256 registers and was 15 % SLOWER than the
// any-party kernel (5.0 vs 4.35 ms per 96 steps x 1024 rotations: +40 % LDS reads per transform at two waves per SIMD).
template <int NP, int L, int PARTY, int WV, bool MARGIN, bool ACCL>
__device__ __forceinline__ void g2_party_steps(const MkGenArgs &P, const int32_t *bara, int32_t *acc, cplx *xch_own, cplx *xch_oth,
                                               const cplx *tw2_lds, const cplx (&tw1f)[8], int32_t xormask, double &worst)
{
    constexpr int PER = 2 * L * NP + 2 * L;       // key polys per (party, bit): x[L][NP] | y[L][NP] | c0[L] | c1[L]
    constexpr int N0 = NP / 2, N1 = NP - 1 - N0;  // non-party sources of wave 0 / wave 1 (N0 = ceil((NP-1)/2))
    constexpr int D0raw = ((N1 - N0 + 2) * L + (N1 - N0)) / 2;
    constexpr int D0 = D0raw < 0 ? 0 : D0raw > L ? L : D0raw;          // body digits [0, D0) -> wave 0, [D0, L) -> wave 1
    const int beta = P.g.log2_base;
    int a_next = load_uniform_i32(bara + (size_t)PARTY * P.n) & (2 * kN - 1);
#pragma unroll 1
    for (int j = 0; j < P.n; j++) {                                              // mk_internals.jl:476
        wave_priority_step(PARTY * P.n + j, P.prio_steps);
        const int a = a_next;
        a_next = load_uniform_i32(bara + (size_t)PARTY * P.n + j + 1) & (2 * kN - 1);    // the row ends with barb: in range
        const int lane = lane_id_fresh();      // per-lane addresses are rebuilt every step, not kept (spilled) across the loop
        const cplx *key = P.bk + ((size_t)PARTY * P.n + j) * PER * kM + lane;
        cplx o_party[8], o_body[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { o_party[q] = mk(0.0, 0.0); o_body[q] = mk(0.0, 0.0); }
        static_for<0, NP + 1>([&](auto s_c) {
            constexpr int s = decltype(s_c)::value;
            constexpr bool is_body = (s == NP), is_party = (s == PARTY);
            constexpr int rank = s < PARTY ? s : s - 1;                         // among the non-party masks
            constexpr bool mine = is_body ? true : is_party ? (WV == 1) : ((rank < N0) == (WV == 0));
            constexpr int p_begin = is_body ? (WV ? D0 : 0) : 0, p_end = !mine ? 0 : is_body ? (WV ? L : D0) : L;
            if constexpr (p_begin < p_end) {
                constexpr bool has_self = !is_body && !is_party;
                cplx o_self[8];
                if constexpr (has_self) {
#pragma unroll
                    for (int q = 0; q < 8; q++) o_self[q] = mk(0.0, 0.0);
                }
                int32_t temp[16];
                rotate_poly<16>(lane, a, acc + s * kImg, P.g.offset, xormask, temp);
#pragma unroll 1
                for (int p = p_begin; p < p_end; p++) {
                    const cplx *k_party = key + (size_t)(is_body ? 2 * L * NP + L + p : L * NP + p * NP + s) * kM;   // c1[p] | y[p, s]   -> a'_party
                    const cplx *k_body = key + (size_t)(is_body ? 2 * L * NP + p : p * NP + s) * kM;                 // c0[p] | x[p, s]   -> b'
                    const cplx *k_self = key + (size_t)(L * NP + p * NP + PARTY) * kM;                               // y[p, party]       -> a'_s
                    cplx kpa[8];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) kpa[k2] = k_party[k2 * 64];
                    cplx x[8];
                    load_digits2(temp, p + 1, beta, x);
                    cplx kbo[8];
                    fft_fwd_wave_mid(lane, x, tw1f, tw2_lds, xch_own, [&]() {
#pragma unroll
                        for (int k2 = 0; k2 < 2; k2++) kbo[k2] = k_body[k2 * 64];
                    });
#pragma unroll
                    for (int k2 = 2; k2 < 8; k2++) kbo[k2] = k_body[k2 * 64];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) o_party[k2] = cfma(x[k2], kpa[k2], o_party[k2]);
                    if constexpr (has_self) {
#pragma unroll
                        for (int k2 = 0; k2 < 8; k2++) kpa[k2] = k_self[k2 * 64];
                    }
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) o_body[k2] = cfma(x[k2], kbo[k2], o_body[k2]);
                    if constexpr (has_self) {
#pragma unroll
                        for (int k2 = 0; k2 < 8; k2++) o_self[k2] = cfma(x[k2], kpa[k2], o_self[k2]);
                    }
                }
                if constexpr (has_self) {      // a'_s complete: only source s feeds it, only this wave read acc[s] in this step
                    fft_inv_wave(lane, o_self, tw1f, tw2_lds, xch_own);
                    accumulate_poly<MARGIN, false>(lane, o_self, acc + s * kImg, &worst);
                }
            }
        });
        // hand-off: each wave leaves the partial sum the other one finishes in its OWN transposition buffer
        WAVE_LDS_FENCE();
        if constexpr (WV == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) xch_own[k2 * 64 + lane] = o_body[k2];
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) xch_own[k2 * 64 + lane] = o_party[k2];
        }
        __syncthreads();
        // the inverse transform runs in the OTHER wave's buffer, the one just read (see mk_blind_rotate_kernel_w2)
        if constexpr (WV == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) o_party[k2] = cadd(o_party[k2], xch_oth[k2 * 64 + lane]);
            WAVE_LDS_FENCE();
            fft_inv_wave(lane, o_party, tw1f, tw2_lds, xch_oth);
            accumulate_poly<MARGIN, false>(lane, o_party, acc + PARTY * kImg, &worst);
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) o_body[k2] = cadd(o_body[k2], xch_oth[k2 * 64 + lane]);
            WAVE_LDS_FENCE();
            fft_inv_wave(lane, o_body, tw1f, tw2_lds, xch_oth);
            accumulate_poly<MARGIN, false>(lane, o_body, acc + NP * kImg, &worst);
        }
        // accumulator stores of this step visible to the other wave of the rotation; also ends the use of the LDS hand-off
        // (ACCL: the accumulators are in LDS and the barrier alone orders them)
        if constexpr (!ACCL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
        if constexpr (!ACCL) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// ACCL (round 4): the accumulator images in LDS instead of global memory.  At 4 parties five images are 21.8 KB; with the
// two transposition buffers a rotation takes 40 192 B, a pair of rotations + the twiddle table 81 408 B = 40 of the 2 KB
// granules LDS is handed out in, so two such workgroups (or one of four rotations) fill a CU's 160 KB exactly and the
// chip still holds 1024 rotations.  The step then needs no workgroup-scope fence and no trip to L2 for the accumulators.
// (8 parties: nine images do not fit; the accumulators stay in global memory.)
template <int NP, int L, bool MARGIN = false, int RW = 2, bool ACCL = false>
__global__ __launch_bounds__(128 * RW, 2) void mk_blind_rotate_kernel_g2(MkGenArgs P)
{
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wib = wave_in_block(), wv = wib & 1, rot = wib >> 1;
    cplx *xch_all = reinterpret_cast<cplx *>(smem) + (size_t)rot * 2 * kXchElems;          // [2 waves][kXchElems]
    cplx *tw2_lds = reinterpret_cast<cplx *>(smem) + (size_t)RW * 2 * kXchElems;           // [8][8]
    cplx *xch_own = xch_all + wv * kXchElems, *xch_oth = xch_all + (1 - wv) * kXchElems;
    const size_t w_raw = (size_t)blockIdx.x * RW + rot;
    const bool live = w_raw < (size_t)P.R;                                       // a padding rotation repeats the last one, stores nothing
    const size_t w = live ? w_raw : (size_t)P.R - 1;
    int32_t *acc;                                                                // [NP+1][kImg]
    if constexpr (ACCL) acc = reinterpret_cast<int32_t *>(smem + ((size_t)RW * 2 * kXchElems + 64) * sizeof(cplx)) + (size_t)rot * (NP + 1) * kImg;
    else acc = P.acc + w_raw * (size_t)(NP + 1) * kImg;                          // global memory
    const int32_t *bara = P.bara + w * ((size_t)NP * P.n + 1);
    const int32_t xormask = gadget_xor_mask(L, P.g.log2_base);

    cplx tw1f[8];
    {
        const int lane0 = lane_id();
#pragma unroll
        for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane0];
        if (wib == 0) tw2_lds[lane0] = P.T.tw2[lane0];
        // acc = (0, ..., 0, X^{-barb} * mu)       mk_internals.jl:491-492, 72-79 : the polynomials are shared out by parity
        for (int s = wv; s < NP; s += 2) init_zero_poly(lane0, acc + s * kImg);
        if (wv == (NP & 1)) init_body_poly(lane0, load_uniform_i32(bara + (size_t)NP * P.n) & (2 * kN - 1), P.mu, acc + NP * kImg);
    }
    if constexpr (!ACCL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if constexpr (!ACCL) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    wave_priority_begin(P.prio_steps);
    // party-major double loop (mk_internals.jl:475-476), one instantiation of the steps per (party, wave)
    if (wv == 0) {
        static_for<0, NP>([&](auto pc) { g2_party_steps<NP, L, decltype(pc)::value, 0, MARGIN, ACCL>(P, bara, acc, xch_own, xch_oth, tw2_lds, tw1f, xormask, worst); });
    } else {
        static_for<0, NP>([&](auto pc) { g2_party_steps<NP, L, decltype(pc)::value, 1, MARGIN, ACCL>(P, bara, acc, xch_own, xch_oth, tw2_lds, tw1f, xormask, worst); });
    }
    if (!live) return;
    const int lane_e = lane_id_fresh();
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0, wv == 0 && lane_e == 0);
    // mk_tlwe_extract_sample (mk_internals.jl:88-95): one extracted mask column per party, b = body[0]
    int32_t *ext = P.ext + w * ((size_t)NP * kN + 1);
    for (int c = wv; c < NP; c += 2) extract_mask_poly(lane_e, acc + c * kImg, ext + (size_t)c * kN);
    if (wv == 0 && lane_e == 0) ext[(size_t)NP * kN] = acc[NP * kImg + kMir];
}
