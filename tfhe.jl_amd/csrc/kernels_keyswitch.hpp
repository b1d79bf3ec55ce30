// kernels_keyswitch.hpp — LWE keyswitch kernels (keyswitch.jl:45-80): v1 gather, v3 tiled integer VALU, v4 int8 MFMA.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/tfhe_mi355x.h"
#include "br_core.hpp"

using namespace tfhe;

// keyswitch.jl:45-80, any base / decomposition length / sample size.  One workgroup per (output sample, chunk of 1024 output
// words); thread w owns words w, w + 256, ... of its chunk.
// Input sample = ext[e0] (+ ext[e1] + (0, 2^29) for MUX, gates.jl:174).
struct KsArgs {
    const int32_t *ext;     // [R][in_stride]
    const int32_t *ks;      // [kN][t][base-1][n+1]
    const int32_t *e0;      // [G] index into ext
    const int32_t *e1;      // [G] second index or -1
    const int32_t *dst;     // [G] output gate index (NULL: identity)
    int32_t *out;           // [B][out_stride]
    int32_t n, kN, t, log2_base;
    // generalised addressing (single key: in_stride = kN+1, in_off = 0, in_b = kN, out_stride = n+1, out_off = 0, out_b = n;
    // multi-key party p (mk_internals.jl:397-411): in_off = p*N, in_b = P*N, out_off = p*n, out_b = P*n)
    int32_t in_stride, in_off, in_b, out_stride, out_off, out_b;
    int32_t add_b;          // 1: out[out_b] = ext b (+ MUX constant) - sum; 0 (multi-key party > 0): accumulate into out[out_b]
};

constexpr int KS1_WPT = 4;  // words per thread: a workgroup covers 1024 output words

#ifdef TFHE_EMIT_KEYSWITCH_KERNELS     // (defined by engine_dispatch.hip, the one translation unit that launches them)
__global__ __launch_bounds__(256) void keyswitch_kernel(KsArgs P)
{
    constexpr int WPT = KS1_WPT;
    const int g = blockIdx.x;
    const int tid = threadIdx.x;
    const int n1 = P.n + 1;
    const int w0 = blockIdx.y * (256 * WPT);                                  // first word of this workgroup's chunk
    const int32_t *x0 = P.ext + (size_t)P.e0[g] * P.in_stride;
    const int e1 = P.e1 ? P.e1[g] : -1;
    const int32_t *x1 = e1 >= 0 ? P.ext + (size_t)e1 * P.in_stride : nullptr;
    const int base1 = (1 << P.log2_base) - 1;
    const uint32_t prec_offset = 1u << (32 - (1 + P.log2_base * P.t));       // keyswitch.jl:58

    uint32_t accw[WPT];
#pragma unroll
    for (int u = 0; u < WPT; u++) accw[u] = 0;

    for (int i = 0; i < P.kN; i++) {
        uint32_t ai = (uint32_t)x0[P.in_off + i];
        if (x1) ai += (uint32_t)x1[P.in_off + i];
        const int32_t aibar = (int32_t)(ai + prec_offset);                  // keyswitch.jl:59
        const int32_t *rows_i = P.ks + (size_t)i * P.t * base1 * n1;
        for (int j = 1; j <= P.t; j++) {
            const int d = (aibar >> (32 - j * P.log2_base)) & base1;         // keyswitch.jl:65-67
            if (d != 0) {                                                    // keyswitch.jl:73
                const int32_t *row = rows_i + (size_t)((j - 1) * base1 + (d - 1)) * n1;
#pragma unroll
                for (int u = 0; u < WPT; u++) {
                    const int wd = w0 + tid + u * 256;
                    if (wd < n1) accw[u] -= (uint32_t)row[wd];               // keyswitch.jl:74
                }
            }
        }
    }
    const size_t og = P.dst ? (size_t)P.dst[g] : (size_t)g;
    int32_t *o = P.out + og * P.out_stride;
#pragma unroll
    for (int u = 0; u < WPT; u++) {
        const int wd = w0 + tid + u * 256;
        if (wd < P.n) {
            o[P.out_off + wd] = (int32_t)accw[u];
        } else if (wd == P.n) {
            uint32_t v = accw[u];
            if (P.add_b) {                                                   // keyswitch.jl:50
                v += (uint32_t)x0[P.in_b];
                if (x1) v += (uint32_t)x1[P.in_b] + (1u << 29);              // gates.jl:174
            } else {
                v += (uint32_t)o[P.out_b];                                   // mk_internals.jl:409 (launches are stream-ordered)
            }
            o[P.out_b] = (int32_t)v;
        }
    }
}
#endif  // TFHE_EMIT_KEYSWITCH_KERNELS

// ---- keyswitch v3 ---------------------------------------------------------------------------------
// Work decomposition: (tile of KS3_G samples) x (slice of kN/KS3_SLICES mask words) x (chunk of 512
// output words).  A lane owns 4 consecutive output words (16-byte loads from the row-padded key), a
// wave-uniform digit selects among the three candidate rows with two scalar bit-masks (s_bfe_i32) and
// four vector ops per word, and the slices' partial sums are combined with integer atomics (exact and
// order-independent).  Blocks are numbered so that blocks sharing a slice share an XCD: each XCD's L2
// then holds only its own 1/8 of the key, which is fetched from beyond L2 once.
constexpr int KS3_G = 16;        // samples per block
constexpr int KS3_SLICES = 16;   // slices of the kN mask words (multiple of 8)

struct Ks3Args {
    const int32_t *ext;     // [R][kN+1]
    const int32_t *ksp;     // [kN][t][base-1][stride]  rows padded to a multiple of 4 words
    const int32_t *e0, *e1, *dst;
    int32_t *out;           // [B][out_stride], pre-initialised to (0, ..., 0, b) by ks3_init_kernel
    int32_t n, kN, t, log2_base, stride, G;
    // generalised addressing (single key: in_stride = kN+1, in_off = 0, in_b = kN, out_stride = n+1,
    // out_off = 0, out_b = n; multi-key party p: in_off = p*N, out_off = p*n, out_b = P*n)
    int32_t in_stride, in_off, in_b, out_stride, out_off, out_b;
};

#ifdef TFHE_EMIT_KEYSWITCH_KERNELS
__global__ void ks3_init_kernel(Ks3Args P)
{
    const int g = blockIdx.x;
    const size_t og = P.dst ? (size_t)P.dst[g] : (size_t)g;
    int32_t *o = P.out + og * P.out_stride;
    for (int w = threadIdx.x; w < P.out_b; w += blockDim.x) o[w] = 0;
    if (threadIdx.x == 0) {
        uint32_t b = (uint32_t)P.ext[(size_t)P.e0[g] * P.in_stride + P.in_b];               // keyswitch.jl:50
        const int e1 = P.e1 ? P.e1[g] : -1;
        if (e1 >= 0) b += (uint32_t)P.ext[(size_t)e1 * P.in_stride + P.in_b] + (1u << 29);  // gates.jl:174
        o[P.out_b] = (int32_t)b;
    }
}
#endif  // TFHE_EMIT_KEYSWITCH_KERNELS

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#ifdef TFHE_EMIT_KEYSWITCH_KERNELS
__global__ __launch_bounds__(128, 2) void keyswitch_kernel_v3(Ks3Args P)
{
    constexpr int G = KS3_G;
    constexpr int JH = 4;                         // digit positions per pipeline stage
    __shared__ int32_t abar[G][128];              // slice length <= 128
    const int tid = threadIdx.x;
    // XCD-aware numbering: consecutive block ids go round-robin over the 8 XCDs
    const int lin = blockIdx.x;
    const int xcd = lin & 7;
    const int rest = lin >> 3;
    const int slice = xcd + 8 * (rest % (KS3_SLICES / 8));
    const int tile = rest / (KS3_SLICES / 8);
    const int wchunk = blockIdx.y;                // 512-word chunk of the output
    const int g0 = tile * G;
    const int slen = P.kN / KS3_SLICES;
    const int i0 = slice * slen;
    const int lb = P.log2_base, tl = P.t;
    const int base1 = (1 << lb) - 1;              // == 3 (checked by the launcher)
    const uint32_t prec_offset = 1u << (32 - (1 + lb * tl));                 // keyswitch.jl:58
    const int w0 = wchunk * 512 + tid * 4;        // first of this lane's 4 words
    const bool active = w0 < P.stride;
    const int wl = active ? w0 : 0;

    for (int idx = tid; idx < G * slen; idx += 128) {
        const int g = idx / slen, ii = idx % slen;
        const int gg = min(g0 + g, P.G - 1);
        uint32_t ai = (uint32_t)P.ext[(size_t)P.e0[gg] * P.in_stride + P.in_off + i0 + ii];
        const int e1 = P.e1 ? P.e1[gg] : -1;
        if (e1 >= 0) ai += (uint32_t)P.ext[(size_t)e1 * P.in_stride + P.in_off + i0 + ii];
        abar[g][ii] = (int32_t)(ai + prec_offset);                           // keyswitch.jl:59
    }
    __syncthreads();

    u32x4 acc[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc[g] = (u32x4)(0u);

    const size_t row_words = (size_t)P.stride;
    const int stages = slen * (tl / JH);          // tl is a multiple of JH (checked by the launcher)
    auto load_stage = [&](int st, u32x4 (&r)[JH][3]) {
        const int ii = st / (tl / JH), jh = st % (tl / JH);
        const int32_t *rows = P.ksp + ((size_t)(i0 + ii) * tl + jh * JH) * base1 * row_words + wl;
#pragma unroll
        for (int j = 0; j < JH; j++)
#pragma unroll
            for (int h = 0; h < 3; h++) r[j][h] = *reinterpret_cast<const u32x4 *>(rows + (size_t)(j * 3 + h) * row_words);
    };
    auto compute_stage = [&](int st, const u32x4 (&r)[JH][3]) {
        const int ii = st / (tl / JH), jh = st % (tl / JH);
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int a = __builtin_amdgcn_readfirstlane(abar[g][ii]);
#pragma unroll
            for (int j = 0; j < JH; j++) {
                const int pos = 32 - (jh * JH + j + 1) * lb;                 // digit = bits [pos, pos+1]  keyswitch.jl:65-67
                const uint32_t m0 = (uint32_t)((a << (31 - pos)) >> 31);     // -(bit 0 of the digit)
                const uint32_t m1 = (uint32_t)((a << (30 - pos)) >> 31);     // -(bit 1 of the digit)
                // digit 0 -> 0, 1 -> r1, 2 -> r2, 3 -> r3                    keyswitch.jl:73-75
                const u32x4 t = (r[j][2] & m1) | (r[j][0] & ~m1);
                const u32x4 u = r[j][1] & m1;
                acc[g] -= (t & m0) | (u & ~m0);
            }
        }
    };

    u32x4 ra[JH][3], rb[JH][3];
    load_stage(0, ra);
    for (int st = 0; st < stages; st += 2) {
        if (st + 1 < stages) load_stage(st + 1, rb);
        compute_stage(st, ra);
        if (st + 2 < stages) load_stage(st + 2, ra);
        if (st + 1 < stages) compute_stage(st + 1, rb);
    }

    if (!active) return;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int gg = g0 + g;
        if (gg >= P.G) break;
        const size_t og = P.dst ? (size_t)P.dst[gg] : (size_t)gg;
        int32_t *o = P.out + og * P.out_stride;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (w0 + q < P.n) atomicAdd(reinterpret_cast<unsigned int *>(o + P.out_off + w0 + q), acc[g][q]);
            else if (w0 + q == P.n) atomicAdd(reinterpret_cast<unsigned int *>(o + P.out_b), acc[g][q]);   // mk_internals.jl:409
        }
    }
}
#endif  // TFHE_EMIT_KEYSWITCH_KERNELS

// ---- keyswitch v4: int8 MFMA ------------------------------------------------------------------------
// out[g][w] = b_g [w == n] - sum_{i,j} KS[i][j][d(g,i,j)][w]   (keyswitch.jl:45-80, no row for digit 0)
// cast as an exact integer contraction  C = A x B:
//   A[g][(i, j, hh)] = 1 if d(g,i,j) == hh            one-hot digits, generated in registers
//   B[(i, j, hh)][(plane, w)] = signed byte `plane` of KS[i][j][hh-1][w]   (hh = 0: zero row)
// with value = sum_plane byte_plane * 256^plane (mod 2^32), bytes in [-128, 127], so every int32 partial
// sum is exact (|C| <= kN*t*128 = 2^20) and out = b - sum_plane C_plane << 8*plane (mod 2^32).
// v_mfma_i32_32x32x32_i8: one instruction covers 32 samples x 32 (plane, word) columns x 32 K-slots = one
// mask word i (8 digit positions x 4 digit values).  Only the pairing of A's and B's K-slots matters:
// lane half h, byte 4q+hh <-> (digit position 4h+q, digit value hh) for both operands.
// A wave owns 64 samples x 32 words x 4 planes (128 accumulator registers); the 4 waves of a block take
// 4 sample groups and share the B stream through L1.  Requires base 4 and t = 8.
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x16 __attribute__((ext_vector_type(16)));

struct Ks4Args {
    const int32_t *ext;
    const i32x4 *bmat;      // [kN][wtiles][4 planes][64 lanes] 16-byte B fragments
    const int32_t *e0, *e1, *dst;
    int32_t *out;
    int32_t n, kN, G, wtiles;
    int32_t in_stride, in_off, in_b, out_stride, out_off, out_b;
    int32_t add_b;          // 1: out[out_b] = ext b (+ MUX constant) - sum; 0 (MK party > 0): accumulate into out_b
    const i32x4 *abar_t;    // [kN/4][Gpad] rounded mask words, 4 consecutive words of one sample per element (ks4_digits_kernel)
    int32_t Gpad;           // samples rounded up to a multiple of 64
    int32_t kslices;        // > 1 (small batches): blockIdx.z takes kN/kslices mask words, results combined with
                            // integer atomics into an output pre-initialised to (0, ..., 0, b) by ks3_init_kernel
};

// balanced signed byte `plane` of a 32-bit word: value == sum_p sbyte(value, p) * 256^p (mod 2^32)
__host__ __device__ inline int32_t signed_byte_plane(uint32_t v, int plane)
{
    int32_t s = 0;
    for (int p = 0; p <= plane; p++) {
        const uint32_t u = v & 255u;
        s = u >= 128u ? (int32_t)u - 256 : (int32_t)u;
        v = (v - (uint32_t)s) >> 8;
    }
    return s;
}

#ifdef TFHE_EMIT_KEYPREP_KERNELS       // (defined by engine_keys.hip)
// key preparation: canonical Int32 [kN][8][3][n+1] -> B fragments
__global__ void ks4_prepare_kernel(const int32_t *__restrict__ ks, i32x4 *__restrict__ bmat, int n, int kN, int wtiles)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (i, wtile, plane, lane)
    const size_t total = (size_t)kN * wtiles * 4 * 64;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const int plane = (int)((idx >> 6) & 3);
    const int wt = (int)((idx >> 8) % wtiles);
    const int i = (int)((idx >> 8) / wtiles);
    const int c = lane & 31, h = lane >> 5;
    const int w = wt * 32 + c;
    i32x4 frag;
    for (int q = 0; q < 4; q++) {
        const int j = 4 * h + q;                                            // digit position (0-based)
        uint32_t word = 0;
        for (int hh = 1; hh <= 3; hh++) {
            int32_t sb = 0;
            if (w <= n) sb = signed_byte_plane((uint32_t)ks[(((size_t)i * 8 + j) * 3 + (hh - 1)) * (n + 1) + w], plane);
            word |= ((uint32_t)sb & 255u) << (8 * hh);
        }
        frag[q] = (int32_t)word;
    }
    bmat[idx] = frag;
}
#endif  // TFHE_EMIT_KEYPREP_KERNELS

#ifdef TFHE_EMIT_KEYSWITCH_KERNELS
// Rounded mask words a_i + 2^15 (keyswitch.jl:58-59; MUX: sum of two extracted samples, gates.jl:174) transposed
// from the sample-major extracted rows into abar_t[i/4][sample] = 4 consecutive words: the MFMA kernel then reads
// one aligned, fully coalesced 16 bytes per sample and stage instead of scattered 4-byte loads.
__global__ __launch_bounds__(128) void ks4_digits_kernel(Ks4Args P, i32x4 *__restrict__ abar_t)
{
    __shared__ uint32_t tile[32][129];
    const int tid = threadIdx.x;
    const int g0 = blockIdx.x * 32, i0 = blockIdx.y * 128;
    for (int g = 0; g < 32; g++) {
        const int gg = min(g0 + g, P.G - 1);
        const int32_t *r0 = P.ext + (size_t)P.e0[gg] * P.in_stride + P.in_off;
        const int e1 = P.e1 ? P.e1[gg] : -1;
        uint32_t v = (uint32_t)r0[i0 + tid];
        if (e1 >= 0) v += (uint32_t)P.ext[(size_t)e1 * P.in_stride + P.in_off + i0 + tid];
        tile[g][tid] = v + (1u << 15);
    }
    __syncthreads();
    for (int k = tid; k < 32 * 32; k += 128) {
        const int q = k >> 5, g = k & 31;              // consecutive threads -> consecutive samples
        i32x4 o;
#pragma unroll
        for (int u = 0; u < 4; u++) o[u] = (int32_t)tile[g][4 * q + u];
        abar_t[(size_t)(i0 / 4 + q) * P.Gpad + g0 + g] = o;
    }
}

__global__ __launch_bounds__(256, 2) void keyswitch_kernel_v4(Ks4Args P)
{
    constexpr int MT = 2;
    // B fragments of 4 consecutive mask words, staged once per block (the 4 waves need the same ones):
    // wave `pl` fetches byte-plane pl, everybody reads all four planes back from LDS.  Double-buffered.
    __shared__ i32x4 bst[2][4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int wt = blockIdx.y;
    const int gbase = (blockIdx.x * 4 + wave) * (32 * MT);
    const bool wave_active = gbase < P.G;            // idle waves still help staging and take the barriers

    // this lane's samples (row of the MFMA A operand): abar_t[stage][sample]
    const i32x4 *ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) ap[mt] = P.abar_t + min(gbase + mt * 32 + c, P.Gpad - 1);

    i32x16 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int pl = 0; pl < 4; pl++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][pl][r] = 0;

    const size_t bstep = (size_t)P.wtiles * 4 * 64;                          // fragments per mask word i
    const int i_begin = (int)blockIdx.z * (P.kN / P.kslices), i_end = i_begin + P.kN / P.kslices;
    // this wave's plane of mask word i: bmat[i][wt][plane = wave][lane]
    const i32x4 *bp = P.bmat + (size_t)i_begin * bstep + ((size_t)wt * 4 + wave) * 64 + lane;
    const int stages = (i_end - i_begin) / 4;
    i32x4 pre[4];
#pragma unroll
    for (int ii = 0; ii < 4; ii++) bst[0][ii][wave][lane] = bp[(size_t)ii * bstep];

    // rounded mask words of this lane's samples for one stage (4 words), fetched one stage ahead
    i32x4 a_cur[MT], a_nxt[MT];
    auto load_a = [&](int i4, i32x4 (&a)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a[mt] = ap[mt][(size_t)(i4 >> 2) * P.Gpad];
    };
    load_a(i_begin, a_cur);

    for (int st = 0; st < stages; st++) {
        const bool more = st + 1 < stages;
        if (more) {
#pragma unroll
            for (int ii = 0; ii < 4; ii++) pre[ii] = bp[(size_t)(4 * (st + 1) + ii) * bstep];
            load_a(i_begin + 4 * (st + 1), a_nxt);
        }
        __syncthreads();     // stage st is visible; everybody is done reading the other buffer (stage st - 1)
        if (wave_active) {
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                i32x4 afrag[MT];
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        // digit position j = 4h+q (0-based) occupies bits [30-2j, 31-2j]   keyswitch.jl:65-67
                        const uint32_t d8 = ((uint32_t)a_cur[mt][ii] >> (27 - 2 * (4 * h + q))) & 24u;   // 8 * digit
                        afrag[mt][q] = (int32_t)(1u << d8);                  // one-hot byte
                    }
#pragma unroll
                for (int pl = 0; pl < 4; pl++) {
                    const i32x4 bfrag = bst[st & 1][ii][pl][lane];
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
                        acc[mt][pl] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[mt], bfrag, acc[mt][pl], 0, 0, 0);
                }
            }
        }
        if (more) {
#pragma unroll
            for (int ii = 0; ii < 4; ii++) bst[(st + 1) & 1][ii][wave][lane] = pre[ii];
#pragma unroll
            for (int mt = 0; mt < MT; mt++) a_cur[mt] = a_nxt[mt];
        }
    }
    if (!wave_active) return;

    // epilogue: C layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    const int w = wt * 32 + c;
    if (w > P.n) return;
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int gg = gbase + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (gg >= P.G) continue;
            uint32_t sum = (uint32_t)acc[mt][0][r] + ((uint32_t)acc[mt][1][r] << 8) + ((uint32_t)acc[mt][2][r] << 16) +
                           ((uint32_t)acc[mt][3][r] << 24);
            const size_t og = P.dst ? (size_t)P.dst[gg] : (size_t)gg;
            int32_t *o = P.out + og * P.out_stride;
            if (P.kslices > 1) {                                             // partial sum of one slice
                atomicAdd(reinterpret_cast<unsigned int *>(o + (w < P.n ? P.out_off + w : P.out_b)), 0u - sum);
            } else if (w < P.n) {
                o[P.out_off + w] = (int32_t)(0u - sum);
            } else {                                                         // the b word
                if (P.add_b) {
                    uint32_t b = (uint32_t)P.ext[(size_t)P.e0[gg] * P.in_stride + P.in_b];             // keyswitch.jl:50
                    const int e1 = P.e1 ? P.e1[gg] : -1;
                    if (e1 >= 0) b += (uint32_t)P.ext[(size_t)e1 * P.in_stride + P.in_b] + (1u << 29);   // gates.jl:174
                    o[P.out_b] = (int32_t)(b - sum);
                } else {
                    o[P.out_b] = (int32_t)((uint32_t)o[P.out_b] - sum);      // mk_internals.jl:409 (launches are stream-ordered)
                }
            }
        }
}
#endif  // TFHE_EMIT_KEYSWITCH_KERNELS

