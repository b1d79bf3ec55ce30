// kernels_n2048.hpp — N = 2048 (BASELINE config 4b): blind_rotate_kernel_n2048x, two waves per blind rotation, radix-2 split.
#pragma once
#include "kernels_common.hpp"

// ---- N = 2048: two waves per blind rotation ----------------------------------------------------------
// M = 1024 folded points.  One radix-2 DIF stage is split across the two waves of a 128-thread block:
//   a_j = z_j + z_{j+512}  -> wave 0 -> even frequencies,   b_j = (z_j - z_{j+512}) W_1024^j -> wave 1 -> odd,
// then each wave runs the same 512-point transform as the N = 1024 kernels on its half, MACs its own
// frequencies and inverse-transforms them; then wave 0 recombines output polynomial 0 and wave 1 polynomial 1 (one
// 8 KB hand-off each way, two synchronisations of the pair per step).  With z_j = u_j w^j, w = e^{-i pi/2048}, w^512 = kappa = e^{-i pi/4}, j = t + 64 r:
//   wave 0 pass-A input  x_r = e^{-i pi r/32}  (u + kappa u'),   lane factor w^t             in tw1f
//   wave 1 pass-A input  x_r = e^{-i pi 5r/32} (u - kappa u'),   lane factor w^t W_1024^t    in tw1f
// Every wave decomposes all four coefficient classes it needs (t+64m, m < 32) itself; the rotation of a polynomial is done
// by one wave and handed to the other (blind_rotate_kernel_n2048x below).
constexpr int kN2 = 2048;

__host__ __device__ constexpr double cos_pi32(int k)    // cos(k pi / 32)
{
    constexpr double T[17] = {1.0, 0.99518472667219692873, 0.98078528040323043058, 0.95694033573220882438,
                              0.92387953251128673848, 0.88192126434835504956, 0.83146961230254523567,
                              0.77301045336273699338, 0.70710678118654752440, 0.63439328416364548779,
                              0.55557023301960228867, 0.47139673682599780857, 0.38268343236508983729,
                              0.29028467725446233105, 0.19509032201612833135, 0.09801714032956077016, 0.0};
    const int m = ((k % 64) + 64) % 64;
    return m <= 16 ? T[m] : m <= 32 ? -T[32 - m] : m <= 48 ? -T[m - 32] : T[64 - m];
}
__host__ __device__ constexpr double sin_pi32(int k) { return cos_pi32(k - 16); }


struct Br2048Args {
    DiagArgs diag;
    const int32_t *bara;   // [R][n+1]
    const cplx *bk;        // [n][L][2][2][2 (wave)][8][64]
    int32_t *ext;          // [R][N+1]
    const cplx *tw1f2;     // [2 (wave)][8][64]
    const cplx *tw2;       // [8][8]
    Gadget g;
    int32_t n, mu;
    int32_t R;             // rotations in the batch (workgroups hold several: the last one may be padded)
    int32_t prio_steps;    // see wave_priority_begin
};

// pass-A input from the four coefficient classes of point jj = t + 64 r (values already converted to double)
//   u = lo - i hi (coefficients jj, jj+1024), u' = lo2 - i hi2 (jj+512, jj+1536); sg = +sqrt(1/2) (wave 0) / -sqrt(1/2)
template <int R>
__device__ __forceinline__ cplx fwd_in_2048(double lo, double hi, double s2, double d2, double sg, bool wave1)
{
    // u +- kappa u' = (lo +- (lo2-hi2)/sqrt2) - i (hi +- (lo2+hi2)/sqrt2),  s2 = lo2-hi2, d2 = lo2+hi2
    const double re = lo + sg * s2, im = hi + sg * d2;
    if (R == 0) return mk(re, -im);
    const double cr = wave1 ? cos_pi32(5 * R) : cos_pi32(R), sr = wave1 ? sin_pi32(5 * R) : sin_pi32(R);
    return mk(re * cr - im * sr, -(re * sr + im * cr));          // (re - i im) e^{-i theta}
}

constexpr int kImg2 = kMir + kN2;       // one N = 2048 polynomial in LDS: mirror | coefficients (rotate_sub3<32>)

// ---- twists by multiples of pi/32 in tan form ------------------------------------------------------------------------------------
// e^{-i K pi/32} = (-i)^q cos(phi) (1 - i tan(phi)) with q the multiple of pi/2 nearest to the angle and |phi| = |J| pi/32 <= pi/4: the
// product by (1 - i tan) is two FMAs, (-i)^q is a swap with signs, and the cosine (0.707 .. 1) is left to ride on an FMA that
// follows (the first butterfly of the transform, the FMA that adds the rounding constant) — br_core.hpp's twist_tan for the
// N = 2048 kernel's angles.  Both waves of that kernel (angles R pi/32 and 5 R pi/32), the recombination (4 R pi/32) and the
// untwist (R pi/32, (R + 8) pi/32) draw on ONE set of constants: tan and cos of j pi/32, j = 1 .. 8.
__host__ __device__ constexpr double tan_pi32(int j)    // tan(j pi / 32), 0 <= j <= 8
{
    constexpr double T[9] = {0.0, 0.0984914033571642530797, 0.198912367379658006913, 0.303346683607342391676, 0.414213562373095048818,
                             0.534511135950791641078, 0.668178637919298920047, 0.820678790828660330965, 1.0};
    return T[j];
}
struct Oct32 { int q, J; };      // angle K pi/32 = q pi/2 + J pi/32, |J| <= 8
__host__ __device__ constexpr Oct32 oct32(int K)
{
    const int Km = ((K % 64) + 64) % 64, qq = (Km + 7) / 16;
    return Oct32{qq % 4, Km - 16 * qq};
}
__host__ __device__ constexpr int scale32_index(int K) { const int J = oct32(K).J; return J < 0 ? -J : J; }      // the cosine left behind is cos(index pi/32)
// The constants as OPAQUE scalar values, made once per kernel: a compile-time double that appears with both signs (fma(-t, b, a) here,
// fma(t, a, b) there) is materialised by the compiler as TWO scalar register pairs, +t and -t, so that the two-operand form of the
// FMA (which has no negation modifier) can be used — 15 constants became 60 scalar registers and the kernel, at its limit of 102,
// spilled scalars and was rescheduled for the worse (round 5: 53 ms against 43; round 6: the same until this).  A value the compiler
// cannot see through is negated by the instruction's own source modifier.
struct Tan32 { double t[9], c[9]; };      // t[j] = tan(j pi/32), c[j] = cos(j pi/32), j = 0 .. 8
__device__ __forceinline__ Tan32 load_tan32()
{
    Tan32 k;
    k.t[0] = 0.0; k.c[0] = 1.0; k.t[8] = 1.0;
    static_for<1, 8>([&](auto jc) { constexpr int j = decltype(jc)::value; k.t[j] = opaque_scalar(tan_pi32(j)); k.c[j] = opaque_scalar(cos_pi32(j)); });
    k.c[8] = opaque_scalar(cos_pi32(8));
    return k;
}
// (a - i b) e^{-i K pi/32} / cos(scale32_index(K) pi/32)
template <int K>
__device__ __forceinline__ cplx twist32_tan(double a, double b, const Tan32 &k)
{
    constexpr Oct32 o = oct32(K);
    constexpr int Ja = o.J < 0 ? -o.J : o.J;
    const double t = k.t[Ja];
    double wr, wi;      // (a - i b)(1 - i ts), ts = +-t:  (a - ts b) - i (b + ts a)
    if (Ja == 0) { wr = a; wi = -b; }
    else if (Ja == 8) { if (o.J > 0) { wr = a - b; wi = -(b + a); } else { wr = a + b; wi = a - b; } }
    else if (o.J > 0) { wr = fma_(-t, b, a); wi = -fma_(t, a, b); }
    else { wr = fma_(t, b, a); wi = fma_(t, a, -b); }
    return o.q == 0 ? mk(wr, wi) : o.q == 1 ? mk(wi, -wr) : o.q == 2 ? mk(-wr, -wi) : mk(-wi, wr);      // times (-i)^q
}
// dft8<false> of x[r] = s[r] w[r] (s[0] = 1): the scales ride on the first butterfly — r = 0: one FMA per component instead of an
// addition; r = 1 .. 3: a multiplication, then the FMA.  58 operations (dft8: 52) for 8 points whose twists cost 14 instead of 28.
template <typename S>
__device__ __forceinline__ void dft8_fwd_scaled_in(cplx (&x)[8], S s, double r2 /* sqrt(1/2) */)
{
    cplx a[4], t[4];
    static_for<0, 4>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        const cplx v = R == 0 ? x[0] : mk(x[R].x * s(rc), x[R].y * s(rc));
        const double s4 = s(std::integral_constant<int, R + 4>{});
        a[R] = axpy(v, s4, x[R + 4]);
        t[R] = axpy(v, -s4, x[R + 4]);
    });
    const cplx b1 = mk(t[1].x + t[1].y, t[1].y - t[1].x);     // t1 * (1 - i)   (lacks 1/sqrt(2))
    const cplx b2 = mk(t[2].y, -t[2].x);                      // t2 * (-i)
    const cplx b3 = mk(t[3].y - t[3].x, -(t[3].x + t[3].y));  // t3 * (-1 - i) (lacks 1/sqrt(2))
    {
        const cplx c0 = cadd(a[0], a[2]), c1 = cadd(a[1], a[3]), d0 = csub(a[0], a[2]), e = csub(a[1], a[3]);
        const cplx d1 = mk(e.y, -e.x);
        x[0] = cadd(c0, c1); x[4] = csub(c0, c1); x[2] = cadd(d0, d1); x[6] = csub(d0, d1);
    }
    {
        const cplx c0 = cadd(t[0], b2), d0 = csub(t[0], b2), c1 = cadd(b1, b3), e = csub(b1, b3);
        const cplx d1 = mk(e.y, -e.x);
        x[1] = axpy(c0, r2, c1); x[5] = axpy(c0, -r2, c1); x[3] = axpy(d0, r2, d1); x[7] = axpy(d0, -r2, d1);
    }
}

// forward 512-point transform of this wave's half (after the radix-2 split), x in / spectrum out
__device__ __forceinline__ void fft_fwd_half(int lane, cplx (&x)[8], const cplx (&tw1f)[8], const cplx *tw2_lds, cplx *xch)
{
    fft_fwd_wave(lane, x, tw1f, tw2_lds, xch);
}

// Recombination of the two inverse half-transforms of one N = 2048 output polynomial (alpha: even frequencies' half, beta:
// odd), untwist, round, add into the polynomial image `ap` (mirror included): the inverse of the radix-2 split above.
template <bool MARGIN, bool TAN = false>
__device__ __forceinline__ void finish_2048(int lane, const cplx (&alpha)[8], const cplx (&beta)[8], int32_t *ap, double &worst, const Tan32 &k)
{
    static_for<0, 8>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        const cplx al = alpha[R], be = beta[R];
        if constexpr (TAN) {
            // the same arithmetic with every twist in tan form (twist32_tan): conj(beta) e_r = sB B', p / m = conj(alpha) +- sB B' (the
            // cosine rides on these additions), p c_r = s0 z0, m c_{r+8} = s1 z1 (the cosines ride on the FMAs that add the rounding constant)
            const cplx Bp = twist32_tan<4 * R>(be.x, be.y, k);
            constexpr int iB = scale32_index(4 * R), i0 = scale32_index(R), i1 = scale32_index(R + 8);
            const double sB = k.c[iB], s0 = k.c[i0], s1 = k.c[i1];
            const double pr = iB == 0 ? al.x + Bp.x : fma_(sB, Bp.x, al.x), pi = iB == 0 ? Bp.y - al.y : fma_(sB, Bp.y, -al.y);
            const double mr = iB == 0 ? al.x - Bp.x : fma_(-sB, Bp.x, al.x), mi = iB == 0 ? -al.y - Bp.y : fma_(-sB, Bp.y, -al.y);
            const cplx z0 = twist32_tan<R>(pr, -pi, k), z1 = twist32_tan<R + 8>(mr, -mi, k);
            if (MARGIN) {
                const double f0 = frac_dist(z0.x * s0), f1 = frac_dist(z0.y * s0), f2 = frac_dist(z1.x * s1), f3 = frac_dist(z1.y * s1);
                worst = f0 > worst ? f0 : worst;
                worst = f1 > worst ? f1 : worst;
                worst = f2 > worst ? f2 : worst;
                worst = f3 > worst ? f3 : worst;
            }
            const int jlo = kMir + lane + 64 * R;
            auto rnd = [](double z, double sc, bool unit) { return (uint32_t)(unit ? round_to_torus32(z) : round_scaled_to_torus32(z, sc)); };
            ap[jlo] = (int32_t)((uint32_t)ap[jlo] + rnd(z0.x, s0, i0 == 0));
            ap[jlo + 1024] = (int32_t)((uint32_t)ap[jlo + 1024] + rnd(z0.y, s0, i0 == 0));
            ap[jlo + 512] = (int32_t)((uint32_t)ap[jlo + 512] + rnd(z1.x, s1, i1 == 0));
            const int32_t last = (int32_t)((uint32_t)ap[jlo + 1536] + rnd(z1.y, s1, i1 == 0));
            ap[jlo + 1536] = last;
            if (R == 7) ap[lane] = (int32_t)(0u - (uint32_t)last);      // coefficient N - 64 + lane: the mirror (rotate_sub3)
            return;
        }
        const double er = cos_pi32(4 * R), ei = -sin_pi32(4 * R);               // e_r = e^{-i pi r/8}
        // conj(beta) e_r   (r = 0 and r = 4 written out: without fast-math the products by 0 and 1 are not folded)
        const double br = R == 0 ? be.x : R == 4 ? -be.y : be.x * er + be.y * ei;
        const double bi = R == 0 ? -be.y : R == 4 ? -be.x : be.x * ei - be.y * er;
        // (conj(alpha) + conj(beta) e_r) c_r      -> coefficients jj, jj+1024        conj(alpha) = (al.x, -al.y)
        // (conj(alpha) - conj(beta) e_r) c_{r+8}  -> coefficients jj+512, jj+1536
        const double pr = al.x + br, pi = -al.y + bi, mr = al.x - br, mi = -al.y - bi;
        const double c0r = cos_pi32(R), c0i = -sin_pi32(R), c1r = cos_pi32(R + 8), c1i = -sin_pi32(R + 8);
        const double re0 = R == 0 ? pr : pr * c0r - pi * c0i, im0 = R == 0 ? pi : pr * c0i + pi * c0r;
        const double re1 = mr * c1r - mi * c1i, im1 = mr * c1i + mi * c1r;
        if (MARGIN) {
            const double f0 = frac_dist(re0), f1 = frac_dist(im0), f2 = frac_dist(re1), f3 = frac_dist(im1);
            worst = f0 > worst ? f0 : worst;
            worst = f1 > worst ? f1 : worst;
            worst = f2 > worst ? f2 : worst;
            worst = f3 > worst ? f3 : worst;
        }
        const int jlo = kMir + lane + 64 * R;
        ap[jlo] = (int32_t)((uint32_t)ap[jlo] + (uint32_t)round_to_torus32(re0));
        ap[jlo + 1024] = (int32_t)((uint32_t)ap[jlo + 1024] + (uint32_t)round_to_torus32(im0));
        ap[jlo + 512] = (int32_t)((uint32_t)ap[jlo + 512] + (uint32_t)round_to_torus32(re1));
        const int32_t last = (int32_t)((uint32_t)ap[jlo + 1536] + (uint32_t)round_to_torus32(im1));
        ap[jlo + 1536] = last;
        if (R == 7) ap[lane] = (int32_t)(0u - (uint32_t)last);      // coefficient N - 64 + lane: the mirror (rotate_sub3)
    });
}

template <bool MARGIN>
__device__ __forceinline__ void finish_2048(int lane, const cplx (&alpha)[8], const cplx (&beta)[8], int32_t *ap, double &worst)
{
    finish_2048<MARGIN, false>(lane, alpha, beta, ap, worst, Tan32{});
}

// ---- N = 2048: the blind-rotation kernel (round 4; round 3's blind_rotate_kernel_n2048 is in the history) ------------------
// Two waves per rotation, each computing one half of the frequencies of every transform (above).  Each half needs every
// coefficient of a rotated polynomial, and until round 3 BOTH waves rotated and offset all 32 coefficient classes of both
// accumulator polynomials: 2 x 406 of a wave's ~4950 instructions per step, half of them scalar address arithmetic.
// Here wave c rotates only polynomial c — the one it also updates, so an accumulator polynomial is private to its wave
// and its update needs no barrier —, runs the L transforms of that polynomial's digits, then parks the 32 rotated words
// (8 KB) in its transposition buffer, which is idle at that point; after the barrier it takes the other wave's words from
// the other buffer and KEEPS that buffer for the remaining transforms (the other wave does the same with this one's).
// Then both inverse half-transforms; wave 0 recombines output polynomial 0 and wave 1 polynomial 1 (finish_2048), so each
// hands ONE 8 KB block over, and that hand-off swaps the buffers back.  A wave's LDS operations execute in order, so a
// buffer a wave has just read is free for it to write; the buffer it gave away it does not touch until the next exchange.
// Two hand-offs per step (a barrier when the workgroup is one rotation, a polled word per wave otherwise: pair_signal), one rotation per
// wave, 16 16-byte LDS operations for the exchange.
// RW rotations per workgroup advance in lockstep (the barriers are workgroup-wide): the transformed key of N = 2048 sets
// (124 MB at n = 630, l = 3) does not stay in the 4 MB L2 of an XCD once workgroups drift apart, and rotations that read the
// same key values at the same time share one trip to the Infinity Cache (one / two / four per workgroup: 53.4 / 44.5 / 47.4 ms).
// Measured against the round-3 kernel on one device, config 4b (profiles/r04/r04a_4b.jsonl, r04a_phase.txt): 44.5 vs 44.9 ms;
// in the stamped builds the rotate phase shrinks from 8.9 k to 3.9 k cycles of a 48 k-cycle step and the other phases take
// up most of what it frees — a wave's issue slots were being used by its partner, not idle.
template <int L, bool MARGIN = false, int RW = 2>
__global__ __launch_bounds__(128 * RW, 2) void blind_rotate_kernel_n2048x(Br2048Args P)
{
    constexpr int K1 = 2;
    // Every constant twist in tan form (round 6; twist32_tan): 92 FP64 instructions less per wave and step, 1 - 1.4 % of the launch.  (Kept
    // as a switch for the general kernel's sake, which shares finish_2048; the round-2 form of the forward twist is fwd_in_2048.)
    constexpr bool TAN = true, TANF = true;
    // The first KPN values of the co = 0 key chunk are requested inside the transform, between the store and the load of its
    // second transposition (x[] is dead there; a chunk requested before the transform spills into the loop: 71.7 ms), the
    // rest after the transform: the L2 round trip then overlaps the last radix-8 pass.  Round 3, one device, 4096 rotations of
    // config 4b: 0: 54.2 ms, 1: 49.8, 2: 48.0, 3: 49.9, 4: 49.3-50.1, 6: 48.6-49.2, 8: 49.4-50.4
    constexpr int KPN = 2;
    // Round 6: with several rotations per workgroup the two hand-offs of a step synchronise only the two waves they concern (pair_signal /
    // pair_wait_take) instead of the whole workgroup; the rotations meet at a barrier every kPairSyncEvery steps, which is what keeps them
    // sharing the key's trips through the L1.  Same device, 4096 rotations of config 4b, barrier at every hand-off: 43.2 / 44.1 ms; meeting
    // every 8 / 16 / 32 / 64 / 128 steps: 42.7 / 42.4 / 42.3 – 42.9 / 42.8 / 42.9; never: 43.1 (the rotations drift apart and each fetches its own
    // key); the words with a barrier at every step as well: 44.0 — a hand-off through a polled word costs more than a barrier that all four
    // waves reach together, and less than waiting for the slower rotation twice per step (profiles/r06/r06p_n2048_pair.txt).  One rotation
    // per workgroup: the pair IS the workgroup, the barrier stays.
    constexpr bool PAIR = RW > 1;
    unsigned long long dg_t0 = 0, dg_r0 = 0;
    diag_begin<MARGIN>(dg_t0, dg_r0);
    double worst = 0.0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wib = wave_in_block();
    const int rot = wib >> 1;                                                     // rotation within the workgroup
    constexpr size_t kRotBytes = K1 * kImg2 * 4 + 2 * kXchElems * sizeof(cplx);
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem + rot * kRotBytes);       // [K1][kImg2]
    cplx *xch_all = reinterpret_cast<cplx *>(smem + rot * kRotBytes + K1 * kImg2 * 4);   // [2][kXchElems]
    cplx *tw2_lds = reinterpret_cast<cplx *>(smem + RW * kRotBytes);              // [8][8]
    int *pair_flags = reinterpret_cast<int *>(smem + RW * kRotBytes + 64 * sizeof(cplx)) + rot * 2;      // [RW][2]: see pair_signal
    const bool wave1_0 = ((tid >> 6) & 1) != 0;
    const int wv = wib & 1;                         // scalar copy: buffer and accumulator bases stay in scalar registers
    int32_t *acc_own = acc_lds + wv * kImg2;                                       // wave c owns polynomial c
    const size_t w_raw = (size_t)blockIdx.x * RW + rot;
    const bool live = w_raw < (size_t)P.R;
    const size_t w = live ? w_raw : (size_t)P.R - 1;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.tw1f2[(wave1_0 ? 512 : 0) + q * 64 + lane0];
    if (tid < 64) tw2_lds[tid] = P.tw2[tid];
    if (lane0 == 0) pair_flags[wib & 1] = 0;
    {
        const int barb = bara[P.n] & (2 * kN2 - 1);
        int32_t v[32];
#pragma unroll
        for (int m = 0; m < 32; m++) {
            const int idx = (lane0 + 64 * m + barb) & (2 * kN2 - 1);
            v[m] = !wave1_0 ? 0 : (idx & kN2) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
        store_cur<32>(lane0, v, acc_own);
    }
    __syncthreads();
    STAMP_DECL;

    int a_next = load_uniform_i32(bara) & (2 * kN2 - 1);
    wave_priority_begin(P.prio_steps);
    Tan32 tk;
    if constexpr (TAN || TANF) tk = load_tan32();
    // One copy of the step loop per wave half, chosen ONCE by a scalar branch: the per-half constants of the radix-2 split (the
    // twist angles, the sign of kappa, which block is handed over) are then compile-time constants.  Round 3 selected them per
    // lane (v_cndmask on every constant: faster than scalar branches around every use, 44.6 vs 46.5 ms); with the whole loop
    // duplicated there is nothing to select (blind_rotate_kernel_h2's recombination gained 6 % from the same change).
    auto steps = [&](auto wvc) {
    constexpr int WV = decltype(wvc)::value;
    constexpr bool wave1 = WV != 0;
    constexpr double sg = wave1 ? -0.70710678118654752440 : 0.70710678118654752440;
#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        wave_priority_step(i, P.prio_steps);
        if (PAIR && RW > 1 && (i & (kPairSyncEvery - 1)) == 0) __syncthreads();      // the rotations of a workgroup stay within kPairSyncEvery steps of one another (they share the key's trips through the L1)
        const int a = a_next;
        a_next = load_uniform_i32(bara + i + 1) & (2 * kN2 - 1);
        // (the lane rebuilt per step: what is derived from it is recomputed here instead of living, and being spilled, across
        //  the whole loop)
        const int lane = lane_id_fresh();
        const cplx *key = P.bk + (size_t)i * (L * K1 * K1 * 2 * kM) + wv * kM;      // (scalar; the lane is added per transform)
        // (zeroed, then accumulated: with the first transform peeled so that its products are written, 43.60 against 43.46 ms)
        cplx out[K1][8];
#pragma unroll
        for (int d = 0; d < K1; d++)
#pragma unroll
            for (int q = 0; q < 8; q++) out[d][q] = mk(0.0, 0.0);
        cplx *xch = xch_all + wv * kXchElems;                   // this wave's buffer at the start of a step
        cplx *xch_other = xch_all + (1 - wv) * kXchElems;
        int32_t temp[32];
        rotate_poly<32>(lane, a, acc_own, P.g.offset, xormask, temp);
        STAMP(0);
        static_for<0, 2>([&](auto phc) {
            constexpr int ph = decltype(phc)::value;
            const int c = ph == 0 ? wv : 1 - wv;                // own polynomial first, then the other wave's
#pragma unroll 1
            for (int p = 0; p < L; p++) {
                cplx x[8];
                static_for<0, 8>([&](auto rc) {
                    constexpr int R = decltype(rc)::value;
                    const int32_t lo = digit2(temp[R], p + 1, beta), l2 = digit2(temp[R + 8], p + 1, beta);
                    const int32_t hi = digit2(temp[R + 16], p + 1, beta), h2 = digit2(temp[R + 24], p + 1, beta);
                    if constexpr (TAN) {
                        // (u +- kappa u') e^{-i theta_R} / cos: theta_R = R pi/32 (wave 0) or 5 R pi/32 (wave 1) — twist32_tan; the cosine rides on the first butterfly
                        const double re = fma_(sg, (double)(l2 - h2), (double)lo), im = fma_(sg, (double)(l2 + h2), (double)hi);
                        x[R] = twist32_tan<(wave1 ? 5 : 1) * R>(re, im, tk);
                    } else {
                        x[R] = fwd_in_2048<R>((double)lo, (double)hi, (double)(l2 - h2), (double)(l2 + h2), sg, wave1);
                    }
                });
                STAMP(1);
                const cplx *kp = key + (size_t)(p * K1 + c) * K1 * 2 * kM + lane;
                cplx kv0[8];
                if constexpr (TAN) dft8_fwd_scaled_in(x, [&](auto rc) { return tk.c[scale32_index((wave1 ? 5 : 1) * decltype(rc)::value)]; }, tk.c[8]);
                fft_fwd_wave_mid<false, TAN>(lane, x, tw1f, tw2_lds, xch, [&]() {
#pragma unroll
                    for (int k2 = 0; k2 < KPN; k2++) kv0[k2] = kp[k2 * 64];
                });
#pragma unroll
                for (int k2 = KPN; k2 < 8; k2++) kv0[k2] = kp[k2 * 64];
                STAMP(2);
                {
                    cplx kv1[8];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) kv1[k2] = kp[(size_t)2 * kM + k2 * 64];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[0][k2] = cfma(x[k2], kv0[k2], out[0][k2]);
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[1][k2] = cfma(x[k2], kv1[k2], out[1][k2]);
                }
                STAMP(3);
            }
            if (ph == 0) {
                // park this wave's rotated words, take the other wave's, keep the buffer they came in
                WAVE_LDS_FENCE();
                int4 *mine = reinterpret_cast<int4 *>(xch);
#pragma unroll
                for (int j = 0; j < 8; j++) mine[j * 64 + lane] = make_int4(temp[4 * j], temp[4 * j + 1], temp[4 * j + 2], temp[4 * j + 3]);
                const int4 *theirs = reinterpret_cast<const int4 *>(xch_other);
                auto take = [&]() {
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int4 v = theirs[j * 64 + lane];
                        temp[4 * j] = v.x; temp[4 * j + 1] = v.y; temp[4 * j + 2] = v.z; temp[4 * j + 3] = v.w;
                    }
                };
                if (PAIR) { pair_signal(pair_flags + wv, 2 * i + 1); pair_wait_take(pair_flags + (1 - wv), 2 * i + 1, take); } else { __syncthreads(); take(); }
                WAVE_LDS_FENCE();
                cplx *t = xch; xch = xch_other; xch_other = t;
                STAMP(10);
            }
        });
        STAMP(4);
        fft_inv_wave(lane, out[0], tw1f, tw2_lds, xch);
        fft_inv_wave(lane, out[1], tw1f, tw2_lds, xch);
        STAMP(5);
        WAVE_LDS_FENCE();
        if (wave1) {
#pragma unroll
            for (int r = 0; r < 8; r++) xch[r * 64 + lane] = out[0][r];
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) xch[r * 64 + lane] = out[1][r];
        }
        cplx oth[8];
        auto take2 = [&]() {
#pragma unroll
            for (int r = 0; r < 8; r++) oth[r] = xch_other[r * 64 + lane];
        };
        if (PAIR) { pair_signal(pair_flags + wv, 2 * i + 2); pair_wait_take(pair_flags + (1 - wv), 2 * i + 2, take2); } else { __syncthreads(); STAMP(6); take2(); }
        STAMP(7);
        auto finish = [&](const cplx (&alpha)[8], const cplx (&beta)[8], int32_t *ap) { finish_2048<MARGIN, TANF>(lane, alpha, beta, ap, worst, tk); };
        if (wave1) finish(oth, out[1], acc_own);
        else finish(out[0], oth, acc_own);
        WAVE_LDS_FENCE();       // (no barrier: only this wave reads or writes acc_own, and the buffer just read is this wave's again)
        STAMP(8);
    }
    };
    if (wv) steps(std::integral_constant<int, 1>{});
    else steps(std::integral_constant<int, 0>{});
    STAMP_FLUSH(P.diag, wib);
    __syncthreads();            // extraction reads both polynomials

    if (!live) return;
    const int tid_e = ((wib & 1) << 6) + lane_id_fresh();
    diag_end<MARGIN>(P.diag, w, worst, dg_t0, dg_r0, tid_e == 0);
    int32_t *ext = P.ext + w * (kN2 + 1);
    for (int j = tid_e; j < kN2; j += 128) {
        const int32_t v = acc_lds[kMir + j];
        if (j == 0) ext[0] = v;
        else ext[kN2 - j] = (int32_t)(0u - (uint32_t)v);
    }
    if (tid_e == 0) ext[kN2] = acc_lds[kImg2 + kMir];
}

// (Round 5, measured and removed — commit "Experiment: one wave per rotation at N = 2048": ONE wave running both halves of the
//  radix-2 split one after the other, all four output half-spectra in registers — no barrier, no exchange, every digit extracted
//  once, but 256 VGPRs + 181 AGPRs of spill space and one wave per SIMD with nobody to issue while it waits: 65.9 vs 46.2 ms per
//  4096 rotations of config 4b on one device, 132.0 vs 88.5 at 8192, 14.4 vs 6.8 for a single rotation
//  (profiles/r05/r05h_n2048_one_wave.jsonl).  The two barriers and the exchange of the two-wave kernel cost less than a second
//  wave per SIMD is worth.)
