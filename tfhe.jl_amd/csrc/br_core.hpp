// br_core.hpp — per-lane building blocks of the blind-rotate kernel (gfx950, wave64).
//
// One 64-lane wave owns one TLWE accumulator for all n CMUX steps.  A polynomial of N = 1024
// coefficients is folded into M = 512 complex points (the reference's transform,
// polynomials.jl:106-132); lane t holds points t + 64 r (r = 0..7), i.e. coefficients t + 64 m
// (m = 0..15).  The 512-point FFT is three radix-8 passes in registers with two transposes through
// a padded LDS exchange buffer.  Output order of the forward transform is the fixed permutation
//     lane L, register k2  <->  frequency (L >> 3) + 8 (L & 7) + 64 k2
// which the inverse transform consumes unchanged and the bootstrapping key is stored in.
//
// Everything here is plain C++ (TFHE_HD) so the same lane code can be executed lane by lane on
// the host (tests/host_sim) as well as in the HIP kernels.
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TFHE_HD __host__ __device__ __forceinline__
typedef double2 cplx;
#else
#define TFHE_HD inline
struct cplx { double x, y; };
#endif

namespace tfhe {

constexpr int kWave = 64;
constexpr int kN = 1024;          // polynomial degree handled by this build of the core
constexpr int kM = kN / 2;        // complex points per polynomial
constexpr int kPts = kM / kWave;  // 8 complex points per lane
constexpr int kXchElems = 576;    // padded exchange buffer, in cplx (8 rows x 72)

TFHE_HD cplx mk(double re, double im) { cplx c; c.x = re; c.y = im; return c; }
TFHE_HD cplx cadd(cplx a, cplx b) { return mk(a.x + b.x, a.y + b.y); }
TFHE_HD cplx csub(cplx a, cplx b) { return mk(a.x - b.x, a.y - b.y); }
// a * b
TFHE_HD cplx cmul(cplx a, cplx b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// a * conj(b)
TFHE_HD cplx cmulc(cplx a, cplx b) { return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }
// acc + a * b
TFHE_HD cplx cfma(cplx a, cplx b, cplx acc)
{
    return mk(acc.x + a.x * b.x - a.y * b.y, acc.y + a.x * b.y + a.y * b.x);
}

// 8-point DFT in registers, natural order in and out.  INV = false: e^{-2 pi i rq/8}.
// 52 double-precision operations: the 1/sqrt(2) of the odd outputs rides on the final FMAs.
template <bool INV>
TFHE_HD void dft8(cplx (&x)[8])
{
    const double s = 0.70710678118654752440;
    const cplx a0 = cadd(x[0], x[4]), a1 = cadd(x[1], x[5]), a2 = cadd(x[2], x[6]), a3 = cadd(x[3], x[7]);
    const cplx t0 = csub(x[0], x[4]), t1 = csub(x[1], x[5]), t2 = csub(x[2], x[6]), t3 = csub(x[3], x[7]);
    cplx b1, b2, b3;  // b1, b3 lack their 1/sqrt(2)
    if (!INV) {
        b1 = mk(t1.x + t1.y, t1.y - t1.x);     // t1 * (1 - i)
        b2 = mk(t2.y, -t2.x);                  // t2 * (-i)
        b3 = mk(t3.y - t3.x, -(t3.x + t3.y));  // t3 * (-1 - i)
    } else {
        b1 = mk(t1.x - t1.y, t1.x + t1.y);     // t1 * (1 + i)
        b2 = mk(-t2.y, t2.x);                  // t2 * (+i)
        b3 = mk(-(t3.x + t3.y), t3.x - t3.y);  // t3 * (-1 + i)
    }
    {   // even outputs: DFT4(a)
        const cplx c0 = cadd(a0, a2), c1 = cadd(a1, a3), d0 = csub(a0, a2), e = csub(a1, a3);
        const cplx d1 = INV ? mk(-e.y, e.x) : mk(e.y, -e.x);
        x[0] = cadd(c0, c1); x[4] = csub(c0, c1); x[2] = cadd(d0, d1); x[6] = csub(d0, d1);
    }
    {   // odd outputs: DFT4(b)
        const cplx c0 = cadd(t0, b2), d0 = csub(t0, b2), c1 = cadd(b1, b3), e = csub(b1, b3);
        const cplx d1 = INV ? mk(-e.y, e.x) : mk(e.y, -e.x);
        x[1] = mk(c0.x + s * c1.x, c0.y + s * c1.y);
        x[5] = mk(c0.x - s * c1.x, c0.y - s * c1.y);
        x[3] = mk(d0.x + s * d1.x, d0.y + s * d1.y);
        x[7] = mk(d0.x - s * d1.x, d0.y - s * d1.y);
    }
}

// Twiddle tables (device global memory / host arrays), built by make_tables().
struct Tables {
    const cplx *tw1;    // [8][64]  e^{-2 pi i t q / 512}          (row 0 unused = 1)
    const cplx *tw2;    // [8][8]   e^{-2 pi i t' q2 / 64}
    const cplx *twist;  // [8][64]  e^{-i pi (t + 64 r) / N}       polynomials.jl:53
    const cplx *tw1f;   // [8][64]  e^{-i pi t / N} * e^{-2 pi i t q / 512}   (tw1 with the lane part of the twist folded in)
};
constexpr int kTableElems = 512 + 64 + 512 + 512;

// Fills a host array of kTableElems cplx: tw1 | tw2 | twist | tw1f (long-double sincos).
template <typename LD, typename COS, typename SIN>
inline void fill_tables(cplx *h, COS cosfn, SIN sinfn)
{
    const LD pi = (LD)3.14159265358979323846264338327950288L;
    for (int q = 0; q < 8; q++)
        for (int t = 0; t < 64; t++) {
            const LD a = (LD)-2 * pi * (LD)(t * q) / (LD)512;
            h[q * 64 + t].x = (double)cosfn(a); h[q * 64 + t].y = (double)sinfn(a);
        }
    for (int q = 0; q < 8; q++)
        for (int t = 0; t < 8; t++) {
            const LD a = (LD)-2 * pi * (LD)(t * q) / (LD)64;
            h[512 + q * 8 + t].x = (double)cosfn(a); h[512 + q * 8 + t].y = (double)sinfn(a);
        }
    for (int r = 0; r < 8; r++)
        for (int t = 0; t < 64; t++) {
            const LD a = -pi * (LD)(t + 64 * r) / (LD)kN;
            h[576 + r * 64 + t].x = (double)cosfn(a); h[576 + r * 64 + t].y = (double)sinfn(a);
        }
    for (int q = 0; q < 8; q++)
        for (int t = 0; t < 64; t++) {
            // e^{-i pi t/N} * e^{-2 pi i t q/512} = e^{-i pi t (1 + 4 q) / N}   (N = 1024)
            const LD a = -pi * (LD)(t * (1 + 4 * q)) / (LD)kN;
            h[1088 + q * 64 + t].x = (double)cosfn(a); h[1088 + q * 64 + t].y = (double)sinfn(a);
        }
}
inline Tables tables_from(const cplx *base)
{
    Tables T;
    T.tw1 = base; T.tw2 = base + 512; T.twist = base + 576; T.tw1f = base + 1088;
    return T;
}

// ---- LDS exchange addressing (units of cplx) ---------------------------------------------------
// exchange 1: [q][t] rows of 64 padded to 72; exchange 2: transposes inside 8-lane groups.
TFHE_HD int x1_a(int lane, int q) { return q * 72 + lane; }                          // lane t, reg q
TFHE_HD int x1_b(int lane, int s) { return (lane >> 3) * 72 + (lane & 7) + 8 * s; }  // lane (q',t'), reg s
TFHE_HD int x2_w(int lane, int q2) { return (lane >> 3) * 72 + q2 * 9 + (lane & 7); }
TFHE_HD int x2_r(int lane, int v) { return (lane >> 3) * 72 + (lane & 7) * 9 + v; }

// ---- forward transform, split at the two LDS round trips ---------------------------------------
TFHE_HD void fwd_pass_a(int lane, cplx (&x)[8], const Tables &T)
{
    dft8<false>(x);
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmul(x[q], T.tw1[q * 64 + lane]);
}
TFHE_HD void fwd_pass_b(int lane, cplx (&x)[8], const Tables &T)
{
    dft8<false>(x);
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmul(x[q], T.tw2[q * 8 + (lane & 7)]);
}
TFHE_HD void fwd_pass_c(cplx (&x)[8]) { dft8<false>(x); }

// ---- inverse transform (unnormalised; the 1/M lives in the key spectra) -------------------------
TFHE_HD void inv_pass_c(cplx (&x)[8]) { dft8<true>(x); }
TFHE_HD void inv_pass_b(int lane, cplx (&x)[8], const Tables &T)
{
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmulc(x[q], T.tw2[q * 8 + (lane & 7)]);
    dft8<true>(x);
}
TFHE_HD void inv_pass_a(int lane, cplx (&x)[8], const Tables &T)
{
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmulc(x[q], T.tw1[q * 64 + lane]);
    dft8<true>(x);
}

// exchange helpers on a generic "LDS" pointer
TFHE_HD void x1_store_a(int lane, const cplx (&x)[8], cplx *xch)
{
#pragma unroll
    for (int q = 0; q < 8; q++) xch[x1_a(lane, q)] = x[q];
}
TFHE_HD void x1_load_a(int lane, cplx (&x)[8], const cplx *xch)
{
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int q = o;
        x[q] = xch[x1_a(lane, q)];
    }
}
TFHE_HD void x1_store_b(int lane, const cplx (&x)[8], cplx *xch)
{
#pragma unroll
    for (int s = 0; s < 8; s++) xch[x1_b(lane, s)] = x[s];
}
TFHE_HD void x1_load_b(int lane, cplx (&x)[8], const cplx *xch)
{
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int s = o;
        x[s] = xch[x1_b(lane, s)];
    }
}
TFHE_HD void x2_store(int lane, const cplx (&x)[8], cplx *xch)
{
#pragma unroll
    for (int q = 0; q < 8; q++) xch[x2_w(lane, q)] = x[q];
}
TFHE_HD void x2_load(int lane, cplx (&x)[8], const cplx *xch)
{
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int v = o;
        x[v] = xch[x2_r(lane, v)];
    }
}

// ---- torus / integer pieces ---------------------------------------------------------------------
// Low 32 bits of round(v) for |v| < 2^51: adding 1.5 * 2^52 leaves round(v) in the low mantissa bits
// in two's complement.  Equivalent to polynomials.jl:115-116 (round(Int64, x) then the low 32 bits).
TFHE_HD int32_t round_to_torus32(double v)
{
    const double t = v + 6755399441055744.0;
#if defined(__HIP_DEVICE_COMPILE__)
    return (int32_t)__double2loint(t);
#else
    uint64_t u;
    memcpy(&u, &t, 8);
    return (int32_t)(uint32_t)u;
#endif
}

// the same of v * m in one FMA (one rounding instead of two before the integer is read off)
TFHE_HD int32_t round_scaled_to_torus32(double v, double m)
{
    const double t = __builtin_fma(v, m, 6755399441055744.0);
#if defined(__HIP_DEVICE_COMPILE__)
    return (int32_t)__double2loint(t);
#else
    uint64_t u;
    memcpy(&u, &t, 8);
    return (int32_t)(uint32_t)u;
#endif
}

// Gadget decomposition constants (tgsw.jl:8-21, 99-117).
struct Gadget {
    int32_t offset;     // sum_p 2^(32 - p beta) * 2^(beta-1), wrapped
    int32_t log2_base;  // beta
    int32_t mask;       // 2^beta - 1
    int32_t half;       // 2^(beta-1)
};
TFHE_HD Gadget make_gadget(int l, int log2_base)
{
    Gadget g;
    uint32_t sum = 0;
    for (int p = 1; p <= l; p++) sum += 1u << (32 - p * log2_base);
    g.offset = (int32_t)(sum * (1u << (log2_base - 1)));
    g.log2_base = log2_base;
    g.mask = (int32_t)((1u << log2_base) - 1);
    g.half = (int32_t)(1u << (log2_base - 1));
    return g;
}
// digit p (1-based) of an already offset coefficient (tgsw.jl:115-116)
TFHE_HD int32_t gadget_digit(int32_t c_plus_offset, int p, const Gadget &g)
{
    return ((c_plus_offset >> (32 - p * g.log2_base)) & g.mask) - g.half;
}

// (X^a - 1) * acc for this lane's 16 coefficients, plus the decomposition offset:
//   temp[m] = rot(acc)[t + 64 m] - acc[t + 64 m] + offset      (bootstrap.jl:21, tlwe.jl:88-93)
// acc_lds holds the whole polynomial; cur[] is this lane's copy of its own coefficients.
TFHE_HD void rotate_sub(int lane, int a_mod_2N, const int32_t *acc_lds, const int32_t (&cur)[16],
                        int32_t offset, int32_t (&temp)[16])
{
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int idx = (lane + 64 * m - a_mod_2N) & (2 * kN - 1);
        const int32_t v = acc_lds[idx & (kN - 1)];
        const uint32_t r = (idx & kN) ? 0u - (uint32_t)v : (uint32_t)v;
        temp[m] = (int32_t)(r - (uint32_t)cur[m] + (uint32_t)offset);
    }
}

// digit polynomial p of temp, folded and twisted: x[r] = (d[t+64r] - i d[t+64r+512]) * twist[r][t]
// (polynomials.jl:110)
TFHE_HD void load_digits(int lane, const int32_t (&temp)[16], int p, const Gadget &g, const Tables &T,
                         cplx (&x)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double lo = (double)gadget_digit(temp[r], p, g);
        const double hi = (double)gadget_digit(temp[r + 8], p, g);
        const cplx w = T.twist[r * 64 + lane];
        x[r] = mk(lo * w.x + hi * w.y, lo * w.y - hi * w.x);
    }
}

// fold + twist of a full-range Int32 polynomial (key preparation; same formula as load_digits)
TFHE_HD void load_poly(int lane, const int32_t *poly, const Tables &T, cplx (&x)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double lo = (double)poly[lane + 64 * r];
        const double hi = (double)poly[lane + 64 * r + kM];
        const cplx w = T.twist[r * 64 + lane];
        x[r] = mk(lo * w.x + hi * w.y, lo * w.y - hi * w.x);
    }
}

// after the inverse transform: conj(y) * twist, real -> coefficient t+64r, imag -> t+64r+512
// (polynomials.jl:127-129), rounded and added into the accumulator (bootstrap.jl:22).
TFHE_HD void untwist_add(int lane, const cplx (&y)[8], const Tables &T, int32_t (&acc)[16])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const cplx w = T.twist[r * 64 + lane];
        const double re = y[r].x * w.x + y[r].y * w.y;
        const double im = y[r].x * w.y - y[r].y * w.x;
        acc[r] = (int32_t)((uint32_t)acc[r] + (uint32_t)round_to_torus32(re));
        acc[r + 8] = (int32_t)((uint32_t)acc[r + 8] + (uint32_t)round_to_torus32(im));
    }
}

// Frequency index held by (lane, reg) after the forward transform.
TFHE_HD int freq_of(int lane, int k2) { return (lane >> 3) + 8 * (lane & 7) + 64 * k2; }

// =================================================================================================
// v2 lane code: twiddles resident in registers, the twist split into a lane part (folded into the
// pass-A twiddles, tw1f) and a register part c_r = e^{-i pi r/16} (compile-time constants), digits by
// signed bit-field extract.  Same transform as above up to rounding.
// =================================================================================================
struct LaneTw {
    cplx tw1f[8];   // [q]  e^{-i pi t/N} e^{-2 pi i t q/512}
    cplx tw2[8];    // [q2] e^{-2 pi i t' q2/64}, t' = lane & 7 ([0] unused)
};
TFHE_HD void load_lane_tw(int lane, const Tables &T, LaneTw &w)
{
#pragma unroll
    for (int q = 0; q < 8; q++) {
        w.tw1f[q] = T.tw1f[q * 64 + lane];
        w.tw2[q] = T.tw2[q * 8 + (lane & 7)];
    }
}
// cos(pi r/16), sin(pi r/16)
TFHE_HD constexpr double twc(int r)
{
    return r == 0 ? 1.0 : r == 1 ? 0.98078528040323044913 : r == 2 ? 0.92387953251128675613
         : r == 3 ? 0.83146961230254523708 : r == 4 ? 0.70710678118654752440 : r == 5 ? 0.55557023301960222474
         : r == 6 ? 0.38268343236508977173 : 0.19509032201612826785;
}
TFHE_HD constexpr double tws(int r)
{
    return r == 0 ? 0.0 : r == 1 ? 0.19509032201612826785 : r == 2 ? 0.38268343236508977173
         : r == 3 ? 0.55557023301960222474 : r == 4 ? 0.70710678118654752440 : r == 5 ? 0.83146961230254523708
         : r == 6 ? 0.92387953251128675613 : 0.98078528040323044913;
}

// The register part of the twist in "tan form": e^{-i pi r/16} = twk(r) (1 - i twt(r)) for r <= 4 and, through
// e^{-i pi r/16} = -i e^{+i pi (8 - r)/16}, = twk(r) (twt(r) - i) for r > 4, with twt(r) = tan(pi min(r, 8 - r)/16) and
// twk(r) = cos(pi min(r, 8 - r)/16): two FMAs per point instead of a multiply and an FMA per component, and only three
// tangents and four cosines as constants (they live in scalar registers).  The cosine left behind rides on the FMAs of the
// butterfly that follows (dft8_fwd_tw) or on the FMA that adds the rounding constant (untwist_add2).
TFHE_HD constexpr double twt(int r)
{
    return (r == 1 || r == 7) ? 0.19891236737965800691 : (r == 2 || r == 6) ? 0.41421356237309504880
         : (r == 3 || r == 5) ? 0.66817863791929891999 : r == 4 ? 1.0 : 0.0;
}
TFHE_HD constexpr double twk(int r) { return twc(r < 4 ? r : 8 - r); }
TFHE_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// xor mask that turns every beta-bit digit field into its signed (two's complement) form
TFHE_HD int32_t gadget_xor_mask(int l, int log2_base)
{
    uint32_t m = 0;
    for (int p = 1; p <= l; p++) m |= 1u << (32 - p * log2_base + log2_base - 1);
    return (int32_t)m;
}
// temp[m] = ((X^a - 1) acc + offset) ^ xormask for this lane's 16 coefficients
TFHE_HD void rotate_sub2(int lane, int a_mod_2N, const int32_t *acc_lds, const int32_t (&cur)[16],
                         int32_t offset, int32_t xormask, int32_t (&temp)[16])
{
    const int base = (lane - a_mod_2N) & (2 * kN - 1);
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int idx = base + 64 * m;                         // bit 10 (mod 2N) = sign, low 10 bits = position
        const int32_t v = acc_lds[idx & (kN - 1)];
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t sgn = (uint32_t)__builtin_amdgcn_sbfe(idx, 10u, 1u);     // 0 or 0xFFFFFFFF
#else
        const uint32_t sgn = (idx & kN) ? 0xFFFFFFFFu : 0u;
#endif
        // (v ^ sgn) - sgn - cur + offset, grouped so that one add folds the sign correction into the constant part
        const uint32_t w = (uint32_t)offset - (uint32_t)cur[m] - sgn;
        temp[m] = (int32_t)((((uint32_t)v ^ sgn) + w) ^ (uint32_t)xormask);
    }
}
// ---- rotation with wave-uniform signs --------------------------------------------------------------------------
// A blind rotation's exponent a is the same for every lane of a wave.  Split a = 64 qa + ra: coefficient lane + 64 m of
// X^a acc is +-acc[(lane - ra) + 64 b] with b = (m - qa) mod NBLK and the sign bit NBLK of (m - qa) — both the same for
// all lanes — except that a lane with lane < ra reaches below block 0 when b = 0: it wants block NBLK - 1 with the
// opposite sign.  The LDS image of a polynomial therefore carries, in front of its N coefficients, a 64-word "mirror"
// holding -acc[N - 64 + t]: the one address formula (lane - ra + 64 + 64 b words into the image) and the one scalar sign
// then serve every lane.  Per coefficient that is an address add, a subtract, one v_xad_u32 ((v ^ sgn) + k) and the
// digit-sign xor — 4 vector instructions where rotate_sub2 (per-lane wrap and sign) takes 9; block offsets, signs and
// the offset constants live in scalar registers.  The mirror costs one extra LDS store per polynomial update.
constexpr int kMir = 64;                   // words in front of the coefficients
template <int NBLK>                        // NBLK = N / 64: 16 (N = 1024) or 32 (N = 2048)
TFHE_HD void rotate_sub3(int lane, int a_mod_2N /* wave-uniform */, const int32_t *img /* mirror | N coefficients */,
                         const int32_t (&cur)[NBLK], int32_t offset, int32_t xormask, int32_t (&temp)[NBLK])
{
    const int ra = a_mod_2N & 63, qa = a_mod_2N >> 6;
    const int32_t *p = img + (lane - ra + kMir);
#pragma unroll
    for (int m = 0; m < NBLK; m++) {
        const int B = (m - qa) & (2 * NBLK - 1);
        const uint32_t sgn = (B & NBLK) ? 0xFFFFFFFFu : 0u;
        const uint32_t v = (uint32_t)p[(B & (NBLK - 1)) * 64];
#if defined(__HIP_DEVICE_COMPILE__)
        // (left to itself the compiler splits this into xor / sub / add3 / xor with the sign bit extracted twice)
        uint32_t c_s = (uint32_t)offset - sgn;                      // scalar
        asm("" : "+s"(c_s));
        const uint32_t k = c_s - (uint32_t)cur[m];
        uint32_t r;
        asm("v_xad_u32 %0, %1, %2, %3" : "=v"(r) : "v"(v), "s"(sgn), "v"(k));
        temp[m] = (int32_t)(r ^ (uint32_t)xormask);
#else
        const uint32_t k = ((uint32_t)offset - sgn) - (uint32_t)cur[m];
        temp[m] = (int32_t)(((v ^ sgn) + k) ^ (uint32_t)xormask);
#endif
    }
}
// this lane's coefficients lane + 64 m of a polynomial image
template <int NBLK>
TFHE_HD void load_cur(int lane, const int32_t *img, int32_t (&cur)[NBLK])
{
#pragma unroll
    for (int m = 0; m < NBLK; m++) cur[m] = img[kMir + lane + 64 * m];
}
// ... and back, refreshing the mirror (the lane's last coefficient is acc[N - 64 + lane])
template <int NBLK>
TFHE_HD void store_cur(int lane, const int32_t (&acc)[NBLK], int32_t *img)
{
#pragma unroll
    for (int m = 0; m < NBLK; m++) img[kMir + lane + 64 * m] = acc[m];
    img[lane] = (int32_t)(0u - (uint32_t)acc[NBLK - 1]);
}

// digit p (1-based) of a prepared coefficient: signed bit-field extract of bits [32 - p beta, 32 - (p-1) beta)
TFHE_HD int32_t digit2(int32_t t, int p, int log2_base)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sbfe(t, (unsigned)(32 - p * log2_base), (unsigned)log2_base);   // one v_bfe_i32
#else
    return (int32_t)((uint32_t)t << ((p - 1) * log2_base)) >> (32 - log2_base);
#endif
}
// x[r] = (d[t+64r] - i d[t+64r+512]) * e^{-i pi r/16}
TFHE_HD void load_digits2(const int32_t (&temp)[16], int p, int log2_base, cplx (&x)[8])
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const double lo = (double)digit2(temp[r], p, log2_base);
        const double hi = (double)digit2(temp[r + 8], p, log2_base);
        if (r == 0) x[r] = mk(lo, -hi);
        else x[r] = mk(lo * twc(r) - hi * tws(r), -(lo * tws(r) + hi * twc(r)));
    }
}
// tan form of one point: (a - i b) e^{-i pi r/16} / twk(r)
template <int R>
TFHE_HD cplx twist_tan(double a, double b)
{
    if (R == 0) return mk(a, -b);
    if (R == 4) return mk(a - b, -(b + a));
    if (R < 4) return mk(fma_(-twt(R), b, a), -fma_(twt(R), a, b));
    return mk(fma_(twt(R), a, -b), -fma_(twt(R), b, a));
}
// u[r] = load_digits2's x[r] / twk(r): 12 FMAs and 2 additions instead of 28 operations
TFHE_HD void load_digits2t(const int32_t (&temp)[16], int p, int log2_base, cplx (&u)[8])
{
#define TFHE_TW_(R) u[R] = twist_tan<R>((double)digit2(temp[R], p, log2_base), (double)digit2(temp[R + 8], p, log2_base))
    TFHE_TW_(0); TFHE_TW_(1); TFHE_TW_(2); TFHE_TW_(3); TFHE_TW_(4); TFHE_TW_(5); TFHE_TW_(6); TFHE_TW_(7);
#undef TFHE_TW_
}
// dft8<false> of x[r] = twk(r) u[r]: the cosines ride on the butterfly's additions as FMA multipliers (the ratio of the
// cosines of the two operands; what is left over at the outputs is twk(0) = 1).  50 operations (dft8: 52).
constexpr double kTwR0 = 0.70710678118654752440;      // twk(4) / twk(0)
constexpr double kTwR1 = 0.84775906502257351226;      // twk(5) / twk(1) = c3 / c1   (twk(6) / twk(2) = 1)
constexpr double kTwR3 = 1.1795804271032745923;       // twk(7) / twk(3) = c1 / c3
constexpr double kTwG0 = 0.92387953251128675613;      // twk(2) / twk(0)            (twk(3) / twk(1) = kTwR1)
constexpr double kTwL = 0.98078528040323044913;       // twk(1) / twk(0)
constexpr double kTwSL = 0.69351992266107373091;      // that / sqrt(2)
TFHE_HD cplx axpy(cplx a, double m, cplx b) { return mk(fma_(m, b.x, a.x), fma_(m, b.y, a.y)); }      // a + m b
TFHE_HD void dft8_fwd_tw(cplx (&x)[8])
{
    const cplx a0 = axpy(x[0], kTwR0, x[4]), a1 = axpy(x[1], kTwR1, x[5]), a2 = cadd(x[2], x[6]), a3 = axpy(x[3], kTwR3, x[7]);
    const cplx t0 = axpy(x[0], -kTwR0, x[4]), t1 = axpy(x[1], -kTwR1, x[5]), t2 = csub(x[2], x[6]), t3 = axpy(x[3], -kTwR3, x[7]);
    const cplx b1 = mk(t1.x + t1.y, t1.y - t1.x);     // t1 * (1 - i)        (scale c1, lacks 1/sqrt(2))
    const cplx b2 = mk(t2.y, -t2.x);                  // t2 * (-i)           (scale c2)
    const cplx b3 = mk(t3.y - t3.x, -(t3.x + t3.y));  // t3 * (-1 - i)       (scale c3, lacks 1/sqrt(2))
    {
        const cplx c0 = axpy(a0, kTwG0, a2), d0 = axpy(a0, -kTwG0, a2), c1 = axpy(a1, kTwR1, a3), e = axpy(a1, -kTwR1, a3);
        const cplx d1 = mk(e.y, -e.x);
        x[0] = axpy(c0, kTwL, c1); x[4] = axpy(c0, -kTwL, c1); x[2] = axpy(d0, kTwL, d1); x[6] = axpy(d0, -kTwL, d1);
    }
    {
        const cplx c0 = axpy(t0, kTwG0, b2), d0 = axpy(t0, -kTwG0, b2), c1 = axpy(b1, kTwR1, b3), e = axpy(b1, -kTwR1, b3);
        const cplx d1 = mk(e.y, -e.x);
        x[1] = axpy(c0, kTwSL, c1); x[5] = axpy(c0, -kTwSL, c1); x[3] = axpy(d0, kTwSL, d1); x[7] = axpy(d0, -kTwSL, d1);
    }
}
TFHE_HD void fwd2_pass_a(cplx (&x)[8], const LaneTw &w)
{
    dft8<false>(x);
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = cmul(x[q], w.tw1f[q]);
}
TFHE_HD void fwd2_pass_b(cplx (&x)[8], const LaneTw &w)
{
    dft8<false>(x);
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmul(x[q], w.tw2[q]);
}
TFHE_HD void inv2_pass_b(cplx (&x)[8], const LaneTw &w)
{
#pragma unroll
    for (int q = 1; q < 8; q++) x[q] = cmulc(x[q], w.tw2[q]);
    dft8<true>(x);
}
TFHE_HD void inv2_pass_a(cplx (&x)[8], const LaneTw &w)
{
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = cmulc(x[q], w.tw1f[q]);
    dft8<true>(x);
}
// distance of v from the nearest integer (rounding-margin diagnostics; |v| < 2^51)
TFHE_HD double frac_dist(double v)
{
    const double r = (v + 6755399441055744.0) - 6755399441055744.0;   // round to nearest integer
    const double d = v - r;
    return d < 0 ? -d : d;
}

// conj(y) * e^{-i pi r/16}: real -> coefficient t+64r, imag -> t+64r+512; round, add into acc.
// MARGIN: also track the largest distance of a pre-round value from an integer (must stay << 0.5).
// FUSED = false: the cosine as a multiplication of its own (a kernel with no two registers to spare for the rounding constant,
// which the fused form needs in vector registers: the FMA's other constant already takes the one scalar operand)
template <bool MARGIN = false, bool FUSED = true>
TFHE_HD void untwist_add2(const cplx (&y)[8], int32_t (&acc)[16], double *worst = nullptr)
{
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if (FUSED) {
            // tan form (twist_tan of conj(y) = y.x - i y.y): the cosine rides on the FMA that adds the rounding constant
            double zr, zi;
            if (r == 0) { zr = y[r].x; zi = y[r].y; }
            else if (r == 4) { zr = y[r].x - y[r].y; zi = y[r].y + y[r].x; }
            else if (r < 4) { zr = fma_(-twt(r), y[r].y, y[r].x); zi = fma_(twt(r), y[r].x, y[r].y); }
            else { zr = fma_(twt(r), y[r].x, -y[r].y); zi = fma_(twt(r), y[r].y, y[r].x); }
            if (MARGIN) {
                const double a = frac_dist(zr * twk(r)), b = frac_dist(zi * twk(r));
                if (a > *worst) *worst = a;
                if (b > *worst) *worst = b;
            }
            acc[r] = (int32_t)((uint32_t)acc[r] + (uint32_t)round_scaled_to_torus32(zr, twk(r)));
            acc[r + 8] = (int32_t)((uint32_t)acc[r + 8] + (uint32_t)round_scaled_to_torus32(zi, -twk(r)));
        } else {
            double re, im;
            if (r == 0) { re = y[r].x; im = -y[r].y; }
            else { re = y[r].x * twc(r) - y[r].y * tws(r); im = -(y[r].x * tws(r) + y[r].y * twc(r)); }
            if (MARGIN) {
                const double a = frac_dist(re), b = frac_dist(im);
                if (a > *worst) *worst = a;
                if (b > *worst) *worst = b;
            }
            acc[r] = (int32_t)((uint32_t)acc[r] + (uint32_t)round_to_torus32(re));
            acc[r + 8] = (int32_t)((uint32_t)acc[r + 8] + (uint32_t)round_to_torus32(im));
        }
    }
}

}  // namespace tfhe
