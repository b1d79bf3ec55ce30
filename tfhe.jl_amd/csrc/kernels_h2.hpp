// kernels_h2.hpp — blind_rotate_kernel_h2: 4 l waves per blind rotation, every transform split over two waves (single gates, circuit
// levels); its 256-point transforms also serve kernels_n512.hpp.
#pragma once
#include "kernels_common.hpp"

// ---- smallest batches: every transform split over two waves ------------------------------------------
// A lone wave issues FP64 at about half its SIMD's rate, so the latency of a CMUX step is set by the number of
// instructions ONE wave runs back to back; giving every transform its own wave (measured) does not help because the step
// is then one forward + one inverse 512-point transform long.  Here each 512-point transform is split over TWO waves
// by a radix-2 decimation in frequency,
//     even frequencies 2k':  FFT256(a),  a_j = z_j + z_{j+256}          odd 2k'+1:  FFT256(b),  b_j = (z_j - z_{j+256}) w^j,
// and each half is a 256-point transform with FOUR points per lane (four radix-4 passes, three wave-private LDS
// transposes of 4 KB).  A rotation is 4 L waves: wave (p, c, h) rotates and decomposes accumulator polynomial c,
// transforms half h of digit p, multiplies into partial sums of both output components, hands them to the owner
// of (co, h) = wave (0, co, h), which sums, inverse-transforms its half, swaps halves with its partner (h ^ 1) and
// updates half of the coefficients of polynomial co.  Three barriers per step.
// Layouts: lane t, register r <-> point j = t + 64 r (j < 256) on input;
//          lane (q, q2, q3) = 16 q + 4 q2 + q3, register q4 <-> half-spectrum index k' = q + 4 q2 + 16 q3 + 64 q4 on output.
struct H2Tables {
    const cplx *tw1h;   // [2 (h)][4 (q)][64 (t)]   e^{-i pi t/N} * (h ? e^{-2 pi i t/512} : 1) * e^{-2 pi i t q/256}
    const cplx *tw2q;   // [4 (q2)][16 (t1)]        e^{-2 pi i t1 q2/64}
    const cplx *tw3q;   // [4 (q3)][4 (t2)]         e^{-2 pi i t2 q3/16}
};
constexpr int kH2TableElems = 2 * 4 * 64 + 4 * 16 + 4 * 4;
constexpr int kH2Buf = 320;          // cplx per wave: transposition buffer (padded 4 x 80)

template <bool INV>
__device__ __forceinline__ void dft4(cplx (&x)[4])
{
    const cplx a = cadd(x[0], x[2]), b = csub(x[0], x[2]), c = cadd(x[1], x[3]), d = csub(x[1], x[3]);
    const cplx id = INV ? mk(-d.y, d.x) : mk(d.y, -d.x);      // forward: -i d, inverse: +i d
    x[0] = cadd(a, c); x[2] = csub(a, c); x[1] = cadd(b, id); x[3] = csub(b, id);
}

struct H2LaneTw { cplx tw1[4], tw2[4], tw3[4]; };

// dft4<false> of x[r] = S_r x'[r] with S = (1, s1, g0, s1 g1): what a first stage leaves when the register part of the twist is
// applied in tan form (load_digits2t / load_digits4t); the scales ride on the additions as FMAs.
// blind_rotate_kernel_h2: S = (1, c1 [/ sqrt 2], c2, c3 [/ sqrt 2]), c_r = cos(pi r/16); N = 512: S = (1, c2, c4, c2)
struct Dft4Scale { double g0, g1, s1; };
__device__ __forceinline__ void dft4_fwd_scaled(cplx (&x)[4], Dft4Scale k)
{
    const cplx a = axpy(x[0], k.g0, x[2]), b = axpy(x[0], -k.g0, x[2]), c = axpy(x[1], k.g1, x[3]), d = axpy(x[1], -k.g1, x[3]);
    const cplx id = mk(d.y, -d.x);
    x[0] = axpy(a, k.s1, c); x[2] = axpy(a, -k.s1, c); x[1] = axpy(b, k.s1, id); x[3] = axpy(b, -k.s1, id);
}

template <bool SCALED = false>
__device__ __forceinline__ void fft256_fwd(int lane, cplx (&x)[4], const H2LaneTw &w, cplx *tb, Dft4Scale k = Dft4Scale{1.0, 1.0, 1.0})
{
    if (SCALED) dft4_fwd_scaled(x, k); else dft4<false>(x);
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = cmul(x[q], w.tw1[q]);
#pragma unroll
    for (int q = 0; q < 4; q++) tb[q * 64 + lane] = x[q];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = tb[(lane >> 4) * 64 + (lane & 15) + 16 * r];
    WAVE_LDS_FENCE();
    dft4<false>(x);
#pragma unroll
    for (int q = 1; q < 4; q++) x[q] = cmul(x[q], w.tw2[q]);
#pragma unroll
    for (int q = 0; q < 4; q++) tb[(lane >> 4) * 80 + q * 20 + (lane & 15)] = x[q];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = tb[(lane >> 4) * 80 + ((lane >> 2) & 3) * 20 + (lane & 3) + 4 * r];
    WAVE_LDS_FENCE();
    dft4<false>(x);
#pragma unroll
    for (int q = 1; q < 4; q++) x[q] = cmul(x[q], w.tw3[q]);
#pragma unroll
    for (int q = 0; q < 4; q++) tb[(lane >> 2) * 20 + q * 5 + (lane & 3)] = x[q];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int r = 0; r < 4; r++) x[r] = tb[(lane >> 2) * 20 + (lane & 3) * 5 + r];
    WAVE_LDS_FENCE();
    dft4<false>(x);
}

__device__ __forceinline__ void fft256_inv(int lane, cplx (&x)[4], const H2LaneTw &w, cplx *tb)
{
    dft4<true>(x);
#pragma unroll
    for (int r = 0; r < 4; r++) tb[(lane >> 2) * 20 + (lane & 3) * 5 + r] = x[r];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = tb[(lane >> 2) * 20 + q * 5 + (lane & 3)];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int q = 1; q < 4; q++) x[q] = cmulc(x[q], w.tw3[q]);
    dft4<true>(x);
#pragma unroll
    for (int r = 0; r < 4; r++) tb[(lane >> 4) * 80 + ((lane >> 2) & 3) * 20 + (lane & 3) + 4 * r] = x[r];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = tb[(lane >> 4) * 80 + q * 20 + (lane & 15)];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int q = 1; q < 4; q++) x[q] = cmulc(x[q], w.tw2[q]);
    dft4<true>(x);
#pragma unroll
    for (int r = 0; r < 4; r++) tb[(lane >> 4) * 64 + (lane & 15) + 16 * r] = x[r];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = tb[q * 64 + lane];
    WAVE_LDS_FENCE();
#pragma unroll
    for (int q = 0; q < 4; q++) x[q] = cmulc(x[q], w.tw1[q]);
    dft4<true>(x);
}

// Recombination of the two inverse half-transforms of one output polynomial by the owner of half H (own: its half, o: the partner's):
//   g_r8 = (a~_r +- conj(kappa)^r b~_r) conj(c_r8), r8 = r + 4 H; coefficient t + 64 r8 = Re g, + 512: -Im g,
// rounded and added to the coefficients read at rotate time (cur), written back with the mirror block (rotate_sub3).
// H is a template argument: with the half a run-time value both twists were computed and one selected — 8 FP64 operations and
// 8 selects per point instead of 4 and none (1.710 -> 1.613 ms per single gate).
// SC (round 6): instead of writing the updated coefficients to the accumulator image, the owner keeps them in registers (cur8:
// classes R8 and R8 + 8, R8 = R + 4 H) and ADDS each one twice into the buffer the NEXT step's rotation will be read from —
// once negated at its own position, once with the rotation's sign at position + a_next: t_next = (X^a' - 1) acc + offset arrives as
// 16 plain reads per wave instead of 32 reads and the rotation arithmetic in each of the 4 l waves (blind_rotate_kernel_h2).
template <int H, bool MARGIN, bool SC = false>
__device__ __forceinline__ void h2_recombine(int lane, const cplx (&own)[4], const cplx (&o)[4], const int32_t (&cur)[16], int32_t *acc_lds, double &worst,
                                             int32_t (*cur8)[4] = nullptr, uint32_t *t_next = nullptr, int a_next = 0)
{
    const double rs = 0.70710678118654752440;
    static_for<0, 4>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        const cplx al = H ? o[R] : own[R], be = H ? own[R] : o[R];
        cplx kb;                              // conj(kappa)^R * be
        if (R == 0) kb = be;
        else if (R == 1) kb = mk((be.x - be.y) * rs, (be.x + be.y) * rs);
        else if (R == 2) kb = mk(-be.y, be.x);
        else kb = mk(-(be.x + be.y) * rs, (be.x - be.y) * rs);
        const cplx wq = H ? csub(al, kb) : cadd(al, kb);
        constexpr int R8 = R + 4 * H;
        // g = wq conj(c_R8), tan form (br_core.hpp, twist_tan): the cosine rides on the rounding FMA
        double zr, zi;
        if (R8 == 0) { zr = wq.x; zi = wq.y; }
        else if (R8 == 4) { zr = wq.x - wq.y; zi = wq.y + wq.x; }
        else if (R8 < 4) { zr = fma_(-twt(R8), wq.y, wq.x); zi = fma_(twt(R8), wq.x, wq.y); }
        else { zr = fma_(twt(R8), wq.x, -wq.y); zi = fma_(twt(R8), wq.y, wq.x); }
        if (MARGIN) {
            const double fa = frac_dist(zr * twk(R8)), fb = frac_dist(zi * twk(R8));
            worst = fa > worst ? fa : worst;
            worst = fb > worst ? fb : worst;
        }
        const int jlo = lane + 64 * R8;
        if constexpr (SC) {
            const int32_t nlo = (int32_t)((uint32_t)cur8[0][R] + (uint32_t)round_scaled_to_torus32(zr, twk(R8)));
            const int32_t nhi = (int32_t)((uint32_t)cur8[1][R] + (uint32_t)round_scaled_to_torus32(zi, -twk(R8)));
            cur8[0][R] = nlo; cur8[1][R] = nhi;
            auto scatter = [&](int j, uint32_t v) {
                atomicAdd(t_next + j, 0u - v);                                   // - acc[j]
                const int idx = j + a_next;                                      // X^a' acc: position j + a' (mod 2N), sign by bit 10
                const uint32_t m = 0u - (((uint32_t)idx >> 10) & 1u);
                atomicAdd(t_next + (idx & (kN - 1)), (v ^ m) - m);
            };
            scatter(jlo, (uint32_t)nlo);
            scatter(jlo + kM, (uint32_t)nhi);
            return;
        }
        const int32_t clo = cur[R8], chi = cur[R8 + 8];     // read at rotate time; nobody else writes them
        const int32_t nlo = (int32_t)((uint32_t)clo + (uint32_t)round_scaled_to_torus32(zr, twk(R8)));
        const int32_t nhi = (int32_t)((uint32_t)chi + (uint32_t)round_scaled_to_torus32(zi, -twk(R8)));
        acc_lds[kMir + jlo] = nlo;
        acc_lds[kMir + jlo + kM] = nhi;
        if (H == 1 && R == 3) acc_lds[lane] = (int32_t)(0u - (uint32_t)nhi);      // coefficient N - 64 + lane: the mirror (rotate_sub3)
    });
}

// (Measured dead end: letting the two waves of a transform each rotate and decompose only half of the lane's points and
//  swap the twisted points through LDS saves a quarter of the forward instructions but costs a fourth barrier: 1.76 ms
//  against 1.70 ms per gate.  Round 4, also measured and removed: FOUR waves per rotation, wave (c, h) running half h of all
//  L digit transforms of polynomial c side by side, stage by stage, so that one transform's LDS round trip overlaps the other's
//  butterflies — two waves rotate a polynomial instead of 2 L, the digits' partial products are summed in registers, one
//  hand-off per wave instead of L + 1, every wave busy through the whole step on a SIMD of its own; bit-identical, 222
//  registers, and slower: 1.75-1.79 against 1.60-1.62 ms (l = 2), 3.02 against 2.48 ms (l = 3).  A lone wave issues an FP64
//  instruction every ~6 cycles whatever its instruction-level parallelism; the second wave on the SIMD is
//  what fills the gaps, and the forward phase of this kernel has it.  Giving the second digit's wave half of the owner's
//  recombination (it shares the owner's SIMD and idles through the inverse phase): 1.615-1.621 against 1.596-1.628 ms, l = 3:
//  2.45 against 2.47-2.48 ms — inside the spread, not kept.)
template <int L, bool MARGIN = false>
__global__ __launch_bounds__(256 * L, 1) void blind_rotate_kernel_h2(BrArgs P, H2Tables HT)
{
    constexpr int K1 = 2, W = 2 * K1 * L;
    // Round 6: the rotation is SCATTERED by the owners instead of gathered by every wave.  All 4 l waves of a rotation need
    // t = (X^a - 1) acc + offset of their polynomial, whole: until round 5 each of them read the accumulator image twice (its own
    // coefficients and the rotated ones: 32 LDS reads and the rotation arithmetic per wave and step, behind the barrier that
    // follows the owners' write).  Now the owner of a coefficient adds it twice — negated at its position, with the rotation's sign
    // at position + a_next (the next exponent is known a step ahead) — into a buffer that the idle waves of digit 1 have reset to
    // `offset`, with LDS atomics (the two owners of a polynomial hit arbitrary positions), and every wave starts the next step with 16
    // plain reads.  The accumulator itself stays in the owners' registers until the extraction.  Same device, interleaved
    // (profiles/r06/r06l_h2_sc.jsonl): a single gate 1.614 -> 1.537 ms, 16 / 64 / 256 rotations 1.64 / 1.64 / 1.66 -> 1.60 / 1.60 / 1.61.
    constexpr bool SC = true;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_all = reinterpret_cast<int32_t *>(smem);                        // [K1][kImg]
    cplx *tb_all = reinterpret_cast<cplx *>(smem + K1 * kImg * 4);               // [W][kH2Buf]
    cplx *extra_all = tb_all + W * kH2Buf;                                       // [W][256]
    uint32_t *tbuf_all = reinterpret_cast<uint32_t *>(extra_all + W * 256);      // SC: [2 (step parity)][K1][kN] rotated differences + offset
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = wave_in_block();                                              // wave = (p, c, h): owners (p = 0) are waves 0..3, one per SIMD
    const int h = wv & 1, c = (wv >> 1) & 1, p = wv >> 2;
    const bool owner = (p == 0);                                                 // owns half h of output component co = c
    int32_t *acc_lds = acc_all + c * kImg;
    cplx *tb = tb_all + wv * kH2Buf, *extra = extra_all + wv * 256;
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);

    H2LaneTw tw;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        tw.tw1[q] = HT.tw1h[(h * 4 + q) * 64 + lane];
        tw.tw2[q] = HT.tw2q[q * 16 + (lane & 15)];
        tw.tw3[q] = HT.tw3q[q * 4 + (lane & 3)];
    }
    if (wv == 0) init_zero_poly(lane, acc_all);
    else if (wv == 1) init_body_poly(lane, bara[P.n] & (2 * kN - 1), P.mu, acc_all + kImg);
    __syncthreads();

    // this lane's four frequencies f = 2 k' + h, k' = q + 4 q2 + 16 q3 + 64 q4: in the key's (v3) order frequency f sits
    // at element (f >> 6) * 64 + (f & 7) * 8 + ((f >> 3) & 7); f = f0 + 128 q4 keeps f & 63
    const int f0 = 2 * ((lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3)) + h;
    const int koff = (f0 >> 6) * 64 + (f0 & 7) * 8 + ((f0 >> 3) & 7);
    // key polys of transform (p, c): [i][p][c][co][512]; both requested one step ahead of their use
    const cplx *key_own = P.bk + (size_t)((p * K1 + c) * K1 + c) * kM + koff;          // co = c
    const cplx *key_oth = P.bk + (size_t)((p * K1 + c) * K1 + (1 - c)) * kM + koff;    // co = 1 - c
    cplx kown[4], koth[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) { kown[q4] = key_own[q4 * 128]; koth[q4] = key_oth[q4 * 128]; }
    int a_next = load_uniform_i32(bara) & (2 * kN - 1);
    int32_t cur8[2][4];                           // SC, owners: the coefficients they update (classes R + 4 h and R + 4 h + 8), kept across the steps
    if constexpr (SC) {
        if (owner) {
#pragma unroll
            for (int r = 0; r < 4; r++) { cur8[0][r] = acc_lds[kMir + lane + 64 * (r + 4 * h)]; cur8[1][r] = acc_lds[kMir + lane + 64 * (r + 4 * h) + kM]; }
        }
        if (p == 0 && h == 0) {                   // step 0's rotated difference, once, by the rotation the other variant runs every step
            int32_t c0[16], t0[16];
            load_cur<16>(lane, acc_lds, c0);
            rotate_sub3<16>(lane, __builtin_amdgcn_readfirstlane(a_next), acc_lds, c0, P.g.offset, 0, t0);
#pragma unroll
            for (int m = 0; m < 16; m++) tbuf_all[c * kN + lane + 64 * m] = (uint32_t)t0[m];
        }
        __syncthreads();
    }
    STAMP_DECL;
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        // Issue priority through the forward phase, where every SIMD holds an owner and the L - 1 waves of the other digits: the
        // hardware favours the oldest wave — the owner —, which then waits at the barrier below for the others to finish alone at a lone
        // wave's rate.  The other digits' waves go first up to the end of their transform, the owner through its products and hand-off
        // (round 6, same device: single gate 1.558 -> 1.516 ms, 256 gates 1.619 -> 1.518, tfhe_parameters_128 2.377 -> 2.247 and
        // 2.406 -> 2.242; raised later in the step or held longer it LOSES 2 - 5 %: profiles/r06/r06q_h2_prio.txt)
        // (l = 3: the third digit's waves one level above the second's — neutral there; written as two tests because THIS form compiles to the
        //  register allocation that measures 0.7 % (one gate) to 1.6 % (64 gates) faster than `p != 0` at l = 2, on two devices: r06q, block 4)
        if (p == 1) __builtin_amdgcn_s_setprio(1); else if (p == 2) __builtin_amdgcn_s_setprio(2);
        const int a = a_next;
        a_next = load_uniform_i32(bara + i + 1) & (2 * kN - 1);   // bara[n] (= barb) exists: harmless read on the last step
        cplx x[4];
        int32_t cur[16];                          // this lane's coefficients of polynomial c (an owner adds its half back at the end)
        uint32_t *t_cur = tbuf_all + ((i & 1) * K1 + c) * kN, *t_next = tbuf_all + (((i + 1) & 1) * K1 + c) * kN;
        {
            int32_t temp[16];
            if constexpr (SC) {
#pragma unroll
                for (int m = 0; m < 16; m++) temp[m] = (int32_t)(t_cur[lane + 64 * m] ^ (uint32_t)xormask);
                (void)a;
            } else {
                load_cur<16>(lane, acc_lds, cur);
                rotate_sub3<16>(lane, __builtin_amdgcn_readfirstlane(a), acc_lds, cur, P.g.offset, xormask, temp);
            }
            STAMP(0);
            // z_r = (d[t+64r] - i d[t+64r+512]) e^{-i pi r/16} = c_r u_r, r < 8 (tan form: load_digits2t); the half's input is
            // z_r + z_{r+4} (h = 0) or (z_r - z_{r+4}) kappa^r, kappa = e^{-i pi/4} (h = 1) = c_r (u_r +- (c_{r+4} / c_r) u_{r+4}) [kappa^r]:
            // the c_r and kappa's 1/sqrt(2) ride on the first butterfly of the transform (dft4_fwd_scaled)
            cplx u[8];
            load_digits2t(temp, p + 1, beta, u);
            if (h == 0) {
                x[0] = axpy(u[0], kTwR0, u[4]); x[1] = axpy(u[1], kTwR1, u[5]); x[2] = cadd(u[2], u[6]); x[3] = axpy(u[3], kTwR3, u[7]);
            } else {
                const cplx d1 = axpy(u[1], -kTwR1, u[5]), d2 = csub(u[2], u[6]), d3 = axpy(u[3], -kTwR3, u[7]);
                x[0] = axpy(u[0], -kTwR0, u[4]);
                x[1] = mk(d1.x + d1.y, d1.y - d1.x);
                x[2] = mk(d2.y, -d2.x);
                x[3] = mk(d3.y - d3.x, -(d3.x + d3.y));
            }
        }
        STAMP(1);
        fft256_fwd<true>(lane, x, tw, tb, Dft4Scale{kTwG0, kTwR1, h ? kTwSL : kTwL});      // s1 = c1 / sqrt(2) : c1
        if (p != 0) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(1);
        STAMP(2);
        cplx own[4], oth[4];                     // this wave's contribution to output component c / 1 - c (half h)
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) { own[q4] = cmul(x[q4], kown[q4]); oth[q4] = cmul(x[q4], koth[q4]); }
        {   // next step's key values (the last step re-reads its own)
            const size_t step = (size_t)(i + 1 < P.n ? i + 1 : i) * (L * K1 * K1 * kM);
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) { kown[q4] = key_own[step + q4 * 128]; koth[q4] = key_oth[step + q4 * 128]; }
        }
        // hand-off: the partial for the OTHER component's output goes to this wave's extra slot; a wave that owns nothing
        // also leaves the one for its own component's output in its transposition buffer (free between transforms)
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) extra[q4 * 64 + lane] = oth[q4];
        if (!owner) {
#pragma unroll
            for (int q4 = 0; q4 < 4; q4++) tb[q4 * 64 + lane] = own[q4];
        }
        if (p == 0) __builtin_amdgcn_s_setprio(0);
        STAMP(3);
        __syncthreads();
        STAMP(4);
        if constexpr (SC) {
            // the waves of digit 1 idle through the inverse phase: they reset the buffer the owners are about to add into (half h of
            // polynomial c each: 512 words, two 16-byte stores per lane) — ordered before those additions by the next barrier
            if (p == 1) {
                const uint4 init = make_uint4((uint32_t)P.g.offset, (uint32_t)P.g.offset, (uint32_t)P.g.offset, (uint32_t)P.g.offset);
                uint4 *q = reinterpret_cast<uint4 *>(t_next + h * (kN / 2));
                q[lane] = init; q[lane + 64] = init;
            }
        }
        if (owner) {
#pragma unroll
            for (int ow = h; ow < W; ow += 2) {          // the waves of the same half
                if (ow == wv) continue;
                const cplx *src = (((ow >> 1) & 1) == c) ? tb_all + ow * kH2Buf : extra_all + ow * 256;
#pragma unroll
                for (int q4 = 0; q4 < 4; q4++) own[q4] = cadd(own[q4], src[q4 * 64 + lane]);
            }
            STAMP(5);
            fft256_inv(lane, own, tw, tb);       // a~_r (h = 0) / b~_r (h = 1), lane factors already removed
            STAMP(6);
            // swap halves with the partner through the owner's own transposition buffer (nobody else reads it)
#pragma unroll
            for (int r = 0; r < 4; r++) tb[r * 64 + lane] = own[r];
        }
        __syncthreads();
        STAMP(7);
        if (owner) {
            const cplx *ps = tb_all + (wv ^ 1) * kH2Buf;
            cplx o[4];
#pragma unroll
            for (int r = 0; r < 4; r++) o[r] = ps[r * 64 + lane];
            // (one copy of the recombination per half, chosen by a scalar branch: h2_recombine)
            if (h) h2_recombine<1, MARGIN, SC>(lane, own, o, cur, acc_lds, worst, cur8, t_next, a_next);
            else h2_recombine<0, MARGIN, SC>(lane, own, o, cur, acc_lds, worst, cur8, t_next, a_next);
        }
        STAMP(8);
        __syncthreads();     // the updated polynomials are visible to every wave's rotation
        STAMP(9);
    }
    if (wv < 4) STAMP_FLUSH(P.diag, wv);
    if constexpr (SC) {                           // the accumulator as the extraction below reads it
        if (owner) {
#pragma unroll
            for (int r = 0; r < 4; r++) { acc_lds[kMir + lane + 64 * (r + 4 * h)] = cur8[0][r]; acc_lds[kMir + lane + 64 * (r + 4 * h) + kM] = cur8[1][r]; }
        }
        __syncthreads();
    }
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0);
    int32_t *ext = P.ext + w * (kN + 1);
    if (wv == 0) extract_mask_poly(lane, acc_all, ext);
    else if (tid == 64) ext[kN] = acc_all[kImg + kMir];
}
