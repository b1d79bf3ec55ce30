// kernels_common.hpp — what every blind-rotation kernel shares: launch arguments, diagnostics, lane / wave helpers, the tan-form
// twists with opaque constants, rotation / accumulation of one polynomial image in LDS, the 512-point transforms of one wave.
// (bootstrap.jl:19-82, tgsw.jl:99-129, polynomials.jl:106-132, tlwe.jl:55-59)
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

// ---- the 512-point transforms of one wave (every kernel but v3, which has them inline) -----------------------------------
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

// ---- hand-offs between the two waves of a rotation without a workgroup barrier (round 6: blind_rotate_kernel_n2048x, mk_blind_rotate_kernel_w2) ----
// A workgroup-wide barrier at every hand-off makes each rotation of a workgroup wait for the slowest one twice per step; the hand-off itself
// concerns two waves.
// Each wave of the pair owns one LDS word; it stores the number of the hand-off it has just written (after the data: a wave's LDS
// operations execute in order) and polls the partner's word until it shows the same number.  Both waves of a pair run the same
// number of steps and a wave signals hand-off k before it waits for k, so neither can wait for a signal that never comes; the poll is
// bounded all the same: kPairGiveUp polls of ~170 cycles each are ~2.5 ms, a hundred steps of the slowest kernel, where a partner that
// is merely behind arrives within a fraction of one step; a wave that gives up leaves wrong words, which every parity test sees, and a
// launch in which every wait gave up would still end after seconds, not hang the device.
constexpr int kPairGiveUp = 1 << 15;
__device__ __forceinline__ void pair_signal(int *flag_own, int k)
{
    WAVE_LDS_FENCE();
    *reinterpret_cast<volatile int *>(flag_own) = k;
}
// `take()` reads the partner's block.  The word is requested FIRST and the block behind it in the same batch: the LDS serves a wave's
// requests in order, so a block read behind a word that already showed k is the block the partner wrote before it — when the partner
// is there already (the usual case: the two waves do the same work) the hand-off costs one LDS round trip, not two.
template <typename TAKE>
__device__ __forceinline__ void pair_wait_take(const int *flag_other, int k, TAKE &&take)
{
    const int f0 = *reinterpret_cast<const volatile int *>(flag_other);
    WAVE_LDS_FENCE();
    take();
    WAVE_LDS_FENCE();
    if (__builtin_amdgcn_readfirstlane(f0) - k >= 0) return;
    // the partner is behind: poll the word alone (a block re-read per poll would take the LDS from the waves that are working), then read
    for (int spin = 0; spin < kPairGiveUp; spin++) {
        __builtin_amdgcn_s_sleep(1);
        const int f = __builtin_amdgcn_readfirstlane(*reinterpret_cast<const volatile int *>(flag_other));
        if (f - k >= 0) break;
    }
    WAVE_LDS_FENCE();
    take();
    WAVE_LDS_FENCE();
}
__device__ __forceinline__ void pair_wait(const int *flag_other, int k)
{
    pair_wait_take(flag_other, k, []() {});
}
// CMUX steps between the workgroup barriers that keep the rotations of a workgroup within reach of one another's key fetches (they share
// the key's trips through the L1): 8 / 16 / 32 / 64 / 128 / never measured on config 4b: 42.7 / 42.4 / 42.3 – 42.9 / 42.8 / 42.9 / 43.1 ms
constexpr int kPairSyncEvery = 32;
