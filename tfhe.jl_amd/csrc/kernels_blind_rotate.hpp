// kernels_blind_rotate.hpp — blind rotation + sample extraction kernels and bootstrapping-key preparation
// (bootstrap.jl:19-82, tgsw.jl:99-129, polynomials.jl:106-132, tlwe.jl:55-59; mk_internals.jl:348-391,464-495).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/tfhe_mi355x.h"
#include <type_traits>

#include "br_core.hpp"

using namespace tfhe;

// Diagnostics written only by the DIAG instantiations (tfhe_set_option("measure_margin", 1)):
//   margin_bits[w] = bit pattern of the largest |pre-round value - nearest integer| of rotation w (non-negative doubles
//                    order like their bit patterns, so waves combine with an integer atomicMax; zeroed before the launch),
//   clk[2w], clk[2w+1] = s_memtime (shader clock) / s_memrealtime (100 MHz) ticks the workgroup's first wave spent in the
//                    kernel: in-kernel clock = clk[2w] / clk[2w+1] x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
struct DiagArgs {
    unsigned long long *margin_bits;
    unsigned long long *clk;
    unsigned long long *phase;   // TFHE_STAMP builds only: [4 waves][16] shader-clock ticks per phase of workgroup 0
};

// Development aid (make stamp -> lib/libtfhe_mi355x_stamp.so, tools/phase_profile.py): with -DTFHE_STAMP the DIAG
// instantiations of the multi-wave kernels also add up, per wave of workgroup 0, the shader-clock ticks between
// consecutive STAMP(k) marks.  Compiled out of the shipped library.
#ifdef TFHE_STAMP
#define STAMP_DECL unsigned long long st_prev_ = __builtin_amdgcn_s_memtime(), st_acc_[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(k) do { if (MARGIN) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); st_acc_[k] += t_ - st_prev_; st_prev_ = t_; } } while (0)
#define STAMP_FLUSH(diag, wave) do { if (MARGIN && blockIdx.x == 0 && (threadIdx.x & 63) == 0 && (diag).phase) { for (int k_ = 0; k_ < 16; k_++) (diag).phase[(wave) * 16 + k_] = st_acc_[k_]; } } while (0)
#else
#define STAMP_DECL do { } while (0)
#define STAMP(k) do { } while (0)
#define STAMP_FLUSH(diag, wave) do { } while (0)
#endif

template <bool DIAG>
__device__ __forceinline__ void diag_begin(unsigned long long &t0, unsigned long long &r0)
{
    if constexpr (DIAG) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
}
template <bool DIAG>
__device__ __forceinline__ void diag_end(const DiagArgs &d, size_t w, double worst, unsigned long long t0, unsigned long long r0, bool clock_writer = (threadIdx.x == 0))
{
    if constexpr (DIAG) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const double o = __shfl_xor(worst, off);
            worst = o > worst ? o : worst;
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMax(&d.margin_bits[w], (unsigned long long)__double_as_longlong(worst));
            if (clock_writer) {
                d.clk[2 * w] = __builtin_amdgcn_s_memtime() - t0;
                d.clk[2 * w + 1] = __builtin_amdgcn_s_memrealtime() - r0;
            }
        }
    }
}

// Index of this thread's wave within its workgroup as a SCALAR: everything derived from it (the rotation it works on, its
// buffers, which half of a split it takes) then lives in scalar registers and branches on it are scalar branches — the
// compiler cannot see that threadIdx.x >> 6 is wave-uniform and otherwise keeps pointers per lane, masks EXEC around
// wave-uniform ifs and, in the register-bound kernels, spills those per-lane copies.
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
// A blind rotation's exponents are read one step ahead through the SCALAR cache (s_load_dword: constant address space): the
// row was written by the prologue kernel of the same batch call, i.e. before this launch, and the value — the same for
// every lane — then waits in a scalar register instead of a vector register that is live across the whole step.
__device__ __forceinline__ int32_t load_uniform_i32(const int32_t *p)
{
    return *(const __attribute__((address_space(4))) int32_t *)(p);
}

// This thread's lane, recomputed (two instructions) instead of read from the register threadIdx.x arrived in: a kernel that is
// short of registers then need not keep (or spill) that register for the whole launch.
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// ... and a copy the compiler cannot merge with earlier ones (it counts up from an opaque zero): for use inside a loop whose
// body should rebuild its per-lane addresses rather than hold them in registers from before the loop.
__device__ __forceinline__ int lane_id_fresh()
{
    unsigned z = 0;
    asm volatile("" : "+s"(z));
    return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
}

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1 (the body needs I as a constant expression)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// ---- the register part of the N = 1024 twist in tan form with OPAQUE constants ------------------------------------------------------
// br_core.hpp's load_digits2t / dft8_fwd_tw with their constants as scalar values the compiler cannot see through (made once per
// kernel: load_tan16).  A compile-time double that is used with both signs is materialised as two scalar register pairs (+c and -c: the
// two-operand FMA has no negation modifier); in a kernel at its scalar-register limit that means spills and a worse schedule (the
// N = 2048 and multi-key kernels lost 16 - 23 % to it in round 5).  An opaque value is negated by the instruction's source modifier.
__device__ __forceinline__ double opaque_scalar(double v) { asm("" : "+s"(v)); return v; }
struct Tan16 { double t[5], r0, r1, r3, g0, l, sl, c[5]; };      // twt(1 .. 3); kTwR0, kTwR1, kTwR3, kTwG0, kTwL, kTwSL; twk(1 .. 4) (load_tan16<true>)
template <bool WITH_COS = false>
__device__ __forceinline__ Tan16 load_tan16()
{
    Tan16 k;
    k.t[0] = 0.0; k.t[4] = 1.0; k.c[0] = 1.0;
    if (WITH_COS) { k.c[1] = opaque_scalar(twk(1)); k.c[2] = opaque_scalar(twk(2)); k.c[3] = opaque_scalar(twk(3)); k.c[4] = opaque_scalar(twk(4)); }
    k.t[1] = opaque_scalar(twt(1)); k.t[2] = opaque_scalar(twt(2)); k.t[3] = opaque_scalar(twt(3));
    k.r0 = opaque_scalar(kTwR0); k.r1 = opaque_scalar(kTwR1); k.r3 = opaque_scalar(kTwR3);
    k.g0 = opaque_scalar(kTwG0); k.l = opaque_scalar(kTwL); k.sl = opaque_scalar(kTwSL);
    return k;
}
template <int R>
__device__ __forceinline__ cplx twist_tan_o(double a, double b, const Tan16 &k)      // twist_tan<R>
{
    if (R == 0) return mk(a, -b);
    if (R == 4) return mk(a - b, -(b + a));
    const double t = k.t[R < 4 ? R : 8 - R];
    if (R < 4) return mk(fma_(-t, b, a), -fma_(t, a, b));
    return mk(fma_(t, a, -b), -fma_(t, b, a));
}
__device__ __forceinline__ void load_digits2t_o(const int32_t (&temp)[16], int p, int log2_base, cplx (&u)[8], const Tan16 &k)
{
    static_for<0, 8>([&](auto rc) {
        constexpr int R = decltype(rc)::value;
        u[R] = twist_tan_o<R>((double)digit2(temp[R], p, log2_base), (double)digit2(temp[R + 8], p, log2_base), k);
    });
}
__device__ __forceinline__ void dft8_fwd_tw_o(cplx (&x)[8], const Tan16 &k)      // dft8_fwd_tw
{
    const cplx a0 = axpy(x[0], k.r0, x[4]), a1 = axpy(x[1], k.r1, x[5]), a2 = cadd(x[2], x[6]), a3 = axpy(x[3], k.r3, x[7]);
    const cplx t0 = axpy(x[0], -k.r0, x[4]), t1 = axpy(x[1], -k.r1, x[5]), t2 = csub(x[2], x[6]), t3 = axpy(x[3], -k.r3, x[7]);
    const cplx b1 = mk(t1.x + t1.y, t1.y - t1.x);
    const cplx b2 = mk(t2.y, -t2.x);
    const cplx b3 = mk(t3.y - t3.x, -(t3.x + t3.y));
    {
        const cplx c0 = axpy(a0, k.g0, a2), d0 = axpy(a0, -k.g0, a2), c1 = axpy(a1, k.r1, a3), e = axpy(a1, -k.r1, a3);
        const cplx d1 = mk(e.y, -e.x);
        x[0] = axpy(c0, k.l, c1); x[4] = axpy(c0, -k.l, c1); x[2] = axpy(d0, k.l, d1); x[6] = axpy(d0, -k.l, d1);
    }
    {
        const cplx c0 = axpy(t0, k.g0, b2), d0 = axpy(t0, -k.g0, b2), c1 = axpy(b1, k.r1, b3), e = axpy(b1, -k.r1, b3);
        const cplx d1 = mk(e.y, -e.x);
        x[1] = axpy(c0, k.sl, c1); x[5] = axpy(c0, -k.sl, c1); x[3] = axpy(d0, k.sl, d1); x[7] = axpy(d0, -k.sl, d1);
    }
}

// untwist_add2<MARGIN, true> (br_core.hpp) with the opaque constants: conj(y) e^{-i pi r/16} in tan form, the cosine on the rounding FMA
template <bool MARGIN>
__device__ __forceinline__ void untwist_add2_o(const cplx (&y)[8], int32_t (&acc)[16], double *worst, const Tan16 &k)
{
    static_for<0, 8>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const double t = k.t[r < 4 ? r : 8 - r], c = k.c[r < 4 ? r : 8 - r];
        double zr, zi;
        if (r == 0) { zr = y[r].x; zi = y[r].y; }
        else if (r == 4) { zr = y[r].x - y[r].y; zi = y[r].y + y[r].x; }
        else if (r < 4) { zr = fma_(-t, y[r].y, y[r].x); zi = fma_(t, y[r].x, y[r].y); }
        else { zr = fma_(t, y[r].x, -y[r].y); zi = fma_(t, y[r].y, y[r].x); }
        if (MARGIN) {
            const double a = frac_dist(zr * c), b = frac_dist(zi * c);
            if (a > *worst) *worst = a;
            if (b > *worst) *worst = b;
        }
        if (r == 0) {
            acc[r] = (int32_t)((uint32_t)acc[r] + (uint32_t)round_to_torus32(zr));
            acc[r + 8] = (int32_t)((uint32_t)acc[r + 8] + (uint32_t)round_to_torus32(-zi));
        } else {
            acc[r] = (int32_t)((uint32_t)acc[r] + (uint32_t)round_scaled_to_torus32(zr, c));
            acc[r + 8] = (int32_t)((uint32_t)acc[r + 8] + (uint32_t)round_scaled_to_torus32(zi, -c));
        }
    });
}

struct BrArgs {
    DiagArgs diag;
    const int32_t *bara;  // [R][n+1], barb last
    const cplx *bk;       // [n][L][K1][K1][8][64] spectra, permuted order, scaled by 1/M
    int32_t *ext;         // [R][(K1-1)*N + 1]
    Tables T;
    Gadget g;
    int32_t n;
    int32_t mu;
    int32_t prio_steps;   // a wave lowers its issue priority 3 -> 2 -> 1 -> 0 over its first prio_steps CMUX steps (wave_priority_* below); 0: never
    int32_t R;            // rotations in the batch
    int32_t l;            // decomposition length, read by the instantiations with L = 0 (any l at run time)
    int32_t grp_big, grp_q;   // blind_rotate_kernel_k2<.., 7>: workgroups [0, grp_big) hold grp_q + 1 rotations, the others grp_q
};

// Issue priority by progress.  The SIMD's arbiter favours the OLDER of its two waves: the first-placed wave of a SIMD runs
// at nearly the speed of a lone wave (4.7 ms per rotation in blind_rotate_kernel_v3) and the second-placed one takes 7.0 ms,
// finishing alone; at the end of a launch every SIMD is left with one wave for milliseconds.  Here a wave sets its own
// priority by its progress (s_setprio 3, 2, 1, 0 over the thirds of its first prio_steps steps, 90 % of the rotation by
// default): the wave with more work left is favoured, the two waves of a SIMD stay closer together and the launch ends
// with less lone-wave time.  Same device, 4096 rotations: 12.64 ms against 13.01 without (12.81 with a single level for the
// first 60 %); 2048 rotations: 6.29 against 6.65; N = 2048: 48.0 vs 49.0 ms, 2-party multi-key: 17.7 vs 18.2 ms with a
// single level.  Option br_prio_pct.
__device__ __forceinline__ void wave_priority_begin(int prio_steps)
{
    if (prio_steps > 0) __builtin_amdgcn_s_setprio(3);
}
__device__ __forceinline__ void wave_priority_step(int step, int prio_steps)
{
    if (prio_steps <= 0) return;
    if (step == prio_steps / 3) __builtin_amdgcn_s_setprio(2);
    else if (step == 2 * prio_steps / 3) __builtin_amdgcn_s_setprio(1);
    else if (step == prio_steps) __builtin_amdgcn_s_setprio(0);
}

// Wave-private LDS hand-off: LDS instructions of one wave execute in issue order, so a compiler-level
// fence is all a single-wave workgroup needs between a ds_write and the ds_read of another lane's data.
#define WAVE_LDS_FENCE() asm volatile("" ::: "memory")

// One polynomial of an accumulator in LDS: mirror | N coefficients (rotate_sub3, br_core.hpp).
constexpr int kImg = kMir + kN;

// rotates polynomial image `img` by X^a (a wave-uniform) and subtracts it: temp = ((X^a - 1) acc + offset) ^ xormask.
// `a` is pinned to a scalar register and made opaque so that the per-block offsets / signs are recomputed (on the scalar
// unit) for every polynomial instead of being kept alive across the transforms.
template <int NBLK>
__device__ __forceinline__ void rotate_poly(int lane, int a, const int32_t *img, int32_t offset, int32_t xormask, int32_t (&temp)[NBLK])
{
    int32_t cur[NBLK];
    load_cur<NBLK>(lane, img, cur);
    int a_here = __builtin_amdgcn_readfirstlane(a);
    asm volatile("" : "+s"(a_here));
    rotate_sub3<NBLK>(lane, a_here, img, cur, offset, xormask, temp);
}
// acc += round(untwisted y), image updated in place (mirror included)
template <bool MARGIN, bool FUSED = true>
__device__ __forceinline__ void accumulate_poly(int lane, const cplx (&y)[8], int32_t *img, double *worst)
{
    int32_t accr[16];
    load_cur<16>(lane, img, accr);
    untwist_add2<MARGIN, FUSED>(y, accr, worst);
    store_cur<16>(lane, accr, img);
}
// accum = (0, ..., 0, X^{-barb} * (mu, ..., mu))     bootstrap.jl:54-56,78 ; tlwe.jl:77-81
__device__ __forceinline__ void init_body_poly(int lane, int barb, int32_t mu, int32_t *img)
{
    int32_t b[16];
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int idx = (lane + 64 * m + barb) & (2 * kN - 1);
        b[m] = (idx & kN) ? (int32_t)(0u - (uint32_t)mu) : mu;
    }
    store_cur<16>(lane, b, img);
}
__device__ __forceinline__ void init_zero_poly(int lane, int32_t *img)
{
    int32_t z[16];
#pragma unroll
    for (int m = 0; m < 16; m++) z[m] = 0;
    store_cur<16>(lane, z, img);
}
// tlwe_extract_sample of one mask polynomial (tlwe.jl:55-59): a'[0] = p[0], a'[m] = -p[N-m]
__device__ __forceinline__ void extract_mask_poly(int lane, const int32_t *img, int32_t *ext)
{
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int j = lane + 64 * m;
        const int32_t v = img[kMir + j];
        if (j == 0) ext[0] = v;
        else ext[kN - j] = (int32_t)(0u - (uint32_t)v);
    }
}

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

// `mid()` runs between the store and the load of the second transposition, when x[] is dead: the place to request global
// data (32 registers are free there) that the caller needs right after the transform.
// TW: x[] comes from load_digits2t (the register part of the twist in tan form; its cosines ride on the first butterfly)
// PRE: the caller has run the first radix-8 pass itself (dft8_fwd_scaled_in)
template <bool TW = false, bool PRE = false, typename MID>
__device__ __forceinline__ void fft_fwd_wave_mid(int lane, cplx (&x)[8], const cplx (&tw1f)[8], const cplx *tw2_lds, cplx *xch, MID &&mid)
{
    if (PRE) { } else if (TW) dft8_fwd_tw(x); else dft8<false>(x);
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = cmul(x[q], tw1f[q]);
    x1_store_a(lane, x, xch);
    WAVE_LDS_FENCE();
    x1_load_b(lane, x, xch);
    dft8<false>(x);
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmul(x[q], tw2_lds[q * 8 + (lane & 7)]);
    WAVE_LDS_FENCE();
    x2_store(lane, x, xch);
    WAVE_LDS_FENCE();
    mid();
    WAVE_LDS_FENCE();
    x2_load(lane, x, xch);
    WAVE_LDS_FENCE();
    dft8<false>(x);
}

template <bool TW = false>
__device__ __forceinline__ void fft_fwd_wave(int lane, cplx (&x)[8], const cplx (&tw1f)[8], const cplx *tw2_lds, cplx *xch)
{
    if (TW) dft8_fwd_tw(x); else dft8<false>(x);
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = cmul(x[q], tw1f[q]);
    x1_store_a(lane, x, xch);
    WAVE_LDS_FENCE();
    x1_load_b(lane, x, xch);
    dft8<false>(x);
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmul(x[q], tw2_lds[q * 8 + (lane & 7)]);
    WAVE_LDS_FENCE();
    x2_store(lane, x, xch);
    WAVE_LDS_FENCE();
    x2_load(lane, x, xch);
    WAVE_LDS_FENCE();
    dft8<false>(x);
}

__device__ __forceinline__ void fft_inv_wave(int lane, cplx (&x)[8], const cplx (&tw1f)[8], const cplx *tw2_lds, cplx *xch)
{
    dft8<true>(x);
    x2_store(lane, x, xch);
    WAVE_LDS_FENCE();
    x2_load(lane, x, xch);
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmulc(x[q], tw2_lds[q * 8 + (lane & 7)]);
    dft8<true>(x);
    WAVE_LDS_FENCE();
    x1_store_b(lane, x, xch);
    WAVE_LDS_FENCE();
    x1_load_a(lane, x, xch);
    WAVE_LDS_FENCE();
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = cmulc(x[q], tw1f[q]);
    dft8<true>(x);
}

// ---- multi-key blind rotation, 2 parties, TWO waves per rotation ------------------------------------------------
// BASELINE config 5 is 1024 rotations: with one wave per rotation that is ONE wave per SIMD (a lone wave issues FP64 at
// about half the SIMD's rate) running 12 forward and 3 inverse transforms per step back to back.  Here the two waves of
// a workgroup split the 3 L forward transforms of a step evenly (wave 0: every digit of the party's mask and half the
// digits of the other mask; wave 1: every digit of the body and the other half) and multiply them into their own
// partial sums of the three new polynomials (mk_internals.jl:371-385); the partial sums are handed over through LDS (wave 1 gives the two mask partials to
// wave 0, wave 0 the body partial to wave 1), each owner adds what it receives, inverse-transforms and updates its
// polynomials.  Two barriers per step (hand-off written / accumulator updated).  All 1024 rotations are resident at two waves per SIMD (39.4 KB of LDS per workgroup: the hand-off
// reuses the transposition buffers).  Same words as the any-party kernel (round 3's one-wave 2-party kernel is gone).  L must be even.
template <int L, int PARTY, int WV, bool MARGIN, int TAN>
__device__ __forceinline__ void mk2_party_steps(int lane_in, const MkBrArgs &P, const int32_t *bara, int32_t *acc_lds,
                                                cplx *xch_own, cplx *xch_oth, cplx *extra, const cplx *tw2_lds, const cplx (&tw1f)[8],
                                                int32_t xormask, double &worst, const Tan16 &tk)
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
        __syncthreads();
        STAMP(4);
        if (WV == 0) {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) { out[0][k2] = cadd(out[0][k2], xch_oth[k2 * 64 + lane]); out[1][k2] = cadd(out[1][k2], extra[k2 * 64 + lane]); }
        } else {
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) out[NP][k2] = cadd(out[NP][k2], xch_oth[k2 * 64 + lane]);
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
        __syncthreads();      // the updated accumulator is visible to both waves' rotations of the next step
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
    // acc = (0, ..., 0, X^{-barb} * mu)       mk_internals.jl:491-492, 72-79
    if (wv == 0) { init_zero_poly(lane, acc_lds); init_zero_poly(lane, acc_lds + kImg); }
    else init_body_poly(lane, bara[NP * P.n] & (2 * kN - 1), P.mu, acc_lds + 2 * kImg);
    __syncthreads();
    wave_priority_begin(P.prio_steps);
    Tan16 tk;
    if constexpr (TAN != 0) tk = load_tan16<(TAN == 2)>();
    // party-major double loop (mk_internals.jl:475-476)
    if (wv == 0) {
        mk2_party_steps<L, 0, 0, MARGIN, TAN>(lane, P, bara, acc_lds, xch_own, xch_oth, extra, tw2_lds, tw1f, xormask, worst, tk);
        mk2_party_steps<L, 1, 0, MARGIN, TAN>(lane, P, bara, acc_lds, xch_own, xch_oth, extra, tw2_lds, tw1f, xormask, worst, tk);
    } else {
        mk2_party_steps<L, 0, 1, MARGIN, TAN>(lane, P, bara, acc_lds, xch_own, xch_oth, extra, tw2_lds, tw1f, xormask, worst, tk);
        mk2_party_steps<L, 1, 1, MARGIN, TAN>(lane, P, bara, acc_lds, xch_own, xch_oth, extra, tw2_lds, tw1f, xormask, worst, tk);
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

// ---- N = 2048: two waves per blind rotation ----------------------------------------------------------
// M = 1024 folded points.  One radix-2 DIF stage is split across the two waves of a 128-thread block:
//   a_j = z_j + z_{j+512}  -> wave 0 -> even frequencies,   b_j = (z_j - z_{j+512}) W_1024^j -> wave 1 -> odd,
// then each wave runs the same 512-point transform as the N = 1024 kernels on its half, MACs its own
// frequencies and inverse-transforms them; then wave 0 recombines output polynomial 0 and wave 1 polynomial 1 (one
// 8 KB hand-off each way, two barriers per step).  With z_j = u_j w^j, w = e^{-i pi/2048}, w^512 = kappa = e^{-i pi/4}, j = t + 64 r:
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
// Two barriers per step, one rotation per wave, 16 16-byte LDS operations for the exchange.
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
                __syncthreads();
                const int4 *theirs = reinterpret_cast<const int4 *>(xch_other);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int4 v = theirs[j * 64 + lane];
                    temp[4 * j] = v.x; temp[4 * j + 1] = v.y; temp[4 * j + 2] = v.z; temp[4 * j + 3] = v.w;
                }
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
        __syncthreads();
        STAMP(6);
        cplx oth[8];
#pragma unroll
        for (int r = 0; r < 8; r++) oth[r] = xch_other[r * 64 + lane];
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
