// tfhe_engine.hip — MI355X (gfx950) TFHE gate-bootstrapping engine: kernels, context, C ABI.
//
// Pipeline of one batch call (tfhe_gates_batch*):
//   prologue_kernel       gate affine prologue (gates.jl) + modulus switch (bootstrap.jl:74-75)
//   blind_rotate_kernel   one wave per blind rotation, accumulator resident in LDS/registers for
//                         all n CMUX steps (bootstrap.jl:19-59, tgsw.jl:99-129, polynomials.jl),
//                         fused test-vector init and sample extraction (tlwe.jl:55-59)
//   keyswitch_kernel      digit-gather-subtract keyswitch (keyswitch.jl:45-80), MUX add fused
//   trivial_gates_kernel  NOT / CONSTANT / COPY (gates.jl:76-93)
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/tfhe_mi355x.h"
#include "br_core.hpp"

using namespace tfhe;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

// Per-opcode affine prologue  t = (0, cst) + sx*x + sy*y  [* 2 for XOR/XNOR]   (gates.jl)
struct GateForm {
    int32_t cst;   // constant added to b
    int8_t sx, sy; // +-1 coefficients (after the optional doubling)
    int8_t mul2;   // (x + y) * 2 form (gates.jl:52,64)
    int8_t use_z;  // second operand comes from in2 (MUX second half)
};

__host__ __device__ inline GateForm gate_form(int kind)
{
    // kind: opcode for plain gates; 100 = MUX first half (AND(x,y)), 101 = MUX second half (ANDNY(x,z))
    const int32_t p8 = (int32_t)(1u << 29), p4 = (int32_t)(1u << 30);
    switch (kind) {
    case TFHE_GATE_NAND:  return {p8, -1, -1, 0, 0};
    case TFHE_GATE_OR:    return {p8, 1, 1, 0, 0};
    case TFHE_GATE_AND:   return {-p8, 1, 1, 0, 0};
    case TFHE_GATE_XOR:   return {p4, 1, 1, 1, 0};
    case TFHE_GATE_XNOR:  return {-p4, -1, -1, 1, 0};
    case TFHE_GATE_NOR:   return {-p8, -1, -1, 0, 0};
    case TFHE_GATE_ANDNY: return {-p8, -1, 1, 0, 0};
    case TFHE_GATE_ANDYN: return {-p8, 1, -1, 0, 0};
    case TFHE_GATE_ORNY:  return {p8, -1, 1, 0, 0};
    case TFHE_GATE_ORYN:  return {p8, 1, -1, 0, 0};
    case 100:             return {-p8, 1, 1, 0, 0};   // gates.jl:166
    case 101:             return {-p8, -1, 1, 0, 1};  // gates.jl:170
    default:              return {0, 0, 0, 0, 0};
    }
}

// rot_a[w] / rot_b[w] = rows of the two operands of rotation w (batch mode: the gate index; level mode: wire
// indices), rot_kind[w] = kind (see gate_form); writes bara[w][0..n] (barb last).
__global__ void prologue_kernel(const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                const int32_t *__restrict__ rot_a, const int32_t *__restrict__ rot_b,
                                const uint8_t *__restrict__ rot_kind, int32_t *__restrict__ bara, int n,
                                int log2_2N)
{
    const int w = blockIdx.x;
    const GateForm f = gate_form(rot_kind[w]);
    const int32_t *x = in0 + (size_t)rot_a[w] * (n + 1);
    const int32_t *y = (f.use_z ? in2 : in1) + (size_t)rot_b[w] * (n + 1);
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        uint32_t v;
        if (f.mul2) {
            v = ((uint32_t)x[i] + (uint32_t)y[i]) * 2u;
            if (f.sx < 0) v = 0u - v;
        } else {
            const uint32_t xv = f.sx > 0 ? (uint32_t)x[i] : 0u - (uint32_t)x[i];
            const uint32_t yv = f.sy > 0 ? (uint32_t)y[i] : 0u - (uint32_t)y[i];
            v = xv + yv;
        }
        if (i == n) v += (uint32_t)f.cst;
        // decode_message(v, 2N): numeric-functions.jl:31-34
        const int32_t r = (int32_t)(v + (1u << (32 - log2_2N - 1))) >> (32 - log2_2N);
        bara[(size_t)w * (n + 1) + i] = r;
    }
}

// modulus switch only (tfhe_bootstrap_batch): bara[w][i] = decode_message(in[w][i], 2N)
__global__ void modswitch_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ bara, int n, int log2_2N)
{
    const size_t w = blockIdx.x;
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        const uint32_t v = (uint32_t)in[w * (n + 1) + i];
        bara[w * (n + 1) + i] = (int32_t)(v + (1u << (32 - log2_2N - 1))) >> (32 - log2_2N);
    }
}

struct BrArgs {
    const int32_t *bara;  // [R][n+1], barb last
    const cplx *bk;       // [n][L][K1][K1][8][64] spectra, permuted order, scaled by 1/M
    int32_t *ext;         // [R][(K1-1)*N + 1]
    Tables T;
    Gadget g;
    int32_t n;
    int32_t mu;
};

template <int K1>
__device__ __forceinline__ void store_acc(int lane, const int32_t (&acc)[16], int32_t *acc_lds)
{
#pragma unroll
    for (int m = 0; m < 16; m++) acc_lds[lane + 64 * m] = acc[m];
}

// One wave = one blind rotation + extraction.
template <int L, int K1>
__global__ __launch_bounds__(64) void blind_rotate_kernel(BrArgs P)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem);            // [K1][N]
    cplx *xch = reinterpret_cast<cplx *>(smem + K1 * kN * 4);        // [kXchElems]
    const int lane = threadIdx.x;
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (P.n + 1);

    // accum = (0, ..., 0, X^{-barb} * (mu, ..., mu))     bootstrap.jl:54-56,78 ; tlwe.jl:77-81
    int32_t acc[K1][16];
    {
        const int barb = bara[P.n] & (2 * kN - 1);
#pragma unroll
        for (int c = 0; c < K1 - 1; c++)
#pragma unroll
            for (int m = 0; m < 16; m++) acc[c][m] = 0;
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int idx = (lane + 64 * m + barb) & (2 * kN - 1);
            acc[K1 - 1][m] = (idx & kN) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
#pragma unroll
        for (int c = 0; c < K1; c++) store_acc<K1>(lane, acc[c], acc_lds + c * kN);
    }
    __syncthreads();

    for (int i = 0; i < P.n; i++) {                                   // bootstrap.jl:33
        const int a = bara[i] & (2 * kN - 1);
        if (a == 0) continue;                                         // bootstrap.jl:34
        const cplx *bki = P.bk + (size_t)i * (L * K1 * K1 * kM);
        cplx out[K1][8];
#pragma unroll
        for (int c = 0; c < K1; c++)
#pragma unroll
            for (int q = 0; q < 8; q++) out[c][q] = mk(0.0, 0.0);

#pragma unroll
        for (int c = 0; c < K1; c++) {
            int32_t temp[16];
            rotate_sub(lane, a, acc_lds + c * kN, acc[c], P.g.offset, temp);   // bootstrap.jl:21
#pragma unroll
            for (int p = 1; p <= L; p++) {
                cplx x[8];
                load_digits(lane, temp, p, P.g, P.T, x);                      // tgsw.jl:126-127
                fwd_pass_a(lane, x, P.T);
                x1_store_a(lane, x, xch);
                __syncthreads();
                x1_load_b(lane, x, xch);
                __syncthreads();
                fwd_pass_b(lane, x, P.T);
                x2_store(lane, x, xch);
                __syncthreads();
                x2_load(lane, x, xch);
                __syncthreads();
                fwd_pass_c(x);
                // out[co] += D[p, c] .* BK_i[p, c].a[co]                      tgsw.jl:128
                const cplx *kp = bki + (size_t)((p - 1) * K1 + c) * K1 * kM + lane;
#pragma unroll
                for (int co = 0; co < K1; co++)
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[co][k2] = cfma(x[k2], kp[(co * 8 + k2) * 64], out[co][k2]);
            }
        }
#pragma unroll
        for (int co = 0; co < K1; co++) {                                      // polynomials.jl:119-132
            inv_pass_c(out[co]);
            x2_store(lane, out[co], xch);
            __syncthreads();
            x2_load(lane, out[co], xch);
            __syncthreads();
            inv_pass_b(lane, out[co], P.T);
            x1_store_b(lane, out[co], xch);
            __syncthreads();
            x1_load_a(lane, out[co], xch);
            __syncthreads();
            inv_pass_a(lane, out[co], P.T);
            untwist_add(lane, out[co], P.T, acc[co]);                          // bootstrap.jl:22
            store_acc<K1>(lane, acc[co], acc_lds + co * kN);
        }
        __syncthreads();
    }

    // tlwe_extract_sample (tlwe.jl:55-59): a'[0] = p[0], a'[m] = -p[N-m]; b' = body[0]
    int32_t *ext = P.ext + w * ((K1 - 1) * kN + 1);
#pragma unroll
    for (int c = 0; c < K1 - 1; c++)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int j = lane + 64 * m;
            if (j == 0) ext[c * kN] = acc[c][m];
            else ext[c * kN + kN - j] = (int32_t)(0u - (uint32_t)acc[c][m]);
        }
    if (lane == 0) ext[(K1 - 1) * kN] = acc[K1 - 1][0];
}

// Wave-private LDS hand-off: LDS instructions of one wave execute in issue order, so a compiler-level
// fence is all a single-wave workgroup needs between a ds_write and the ds_read of another lane's data.
#define WAVE_LDS_FENCE() asm volatile("" ::: "memory")

// v3: one wave per blind rotation at 2 waves/SIMD (<= 256 VGPRs, no AGPR/scratch spills).
//   * pass-A twiddles (with the lane part of the twist folded in) resident in registers, pass-B twiddles
//     in a 1 KB wave-private LDS table, the register part of the twist as compile-time constants:
//     no global loads on the critical path except the key;
//   * the accumulator lives only in LDS (read at rotate time and at the final add);
//   * key spectra of the next transform prefetched into registers while the current FFT runs;
//   * no s_barrier: wave-private LDS needs only compiler-level ordering;
//   * no branch on bara[i] == 0 (the step then adds exactly zero).
template <int L, int KPF /* key values prefetched per transform: 16 = whole chunk, 8 = half */, bool TW2REG = false /* pass-B twiddles in registers instead of LDS */>
__global__ __launch_bounds__(64, 2) void blind_rotate_kernel_v3(BrArgs P)
{
    constexpr int K1 = 2;
    constexpr int F = K1 * L;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem);                    // [K1][N]
    cplx *xch = reinterpret_cast<cplx *>(smem + K1 * kN * 4);                // [kXchElems]
    cplx *tw2_lds = xch + kXchElems;                                         // [8][8]
    const int lane = threadIdx.x;
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    tw2_lds[lane] = P.T.tw2[lane];
    cplx tw2r[8];
    if (TW2REG) {
#pragma unroll
        for (int q = 1; q < 8; q++) tw2r[q] = P.T.tw2[q * 8 + (lane & 7)];
    }
    {
        const int barb = bara[P.n] & (2 * kN - 1);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int idx = (lane + 64 * m + barb) & (2 * kN - 1);
            acc_lds[lane + 64 * m] = 0;
            acc_lds[kN + lane + 64 * m] = (idx & kN) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
    }
    WAVE_LDS_FENCE();

    cplx kbuf[16];
    // chunk f of step: key spectra for transform f = (c, p): 16 values per lane (co-major, k2 minor)
    auto key_ptr = [&](int step, int f) {
        const int c = f / L, p = f % L;
        return P.bk + (size_t)step * (L * K1 * K1 * kM) + (size_t)(p * K1 + c) * K1 * kM + lane;
    };
    {
        const cplx *kp = key_ptr(0, 0);
#pragma unroll
        for (int j = 0; j < KPF; j++) kbuf[j] = kp[j * 64];
    }

    int a_next = bara[0] & (2 * kN - 1);
    for (int i = 0; i < P.n; i++) {
        const int a = a_next;
        a_next = bara[i + 1] & (2 * kN - 1);   // bara[n] (= barb) exists: harmless read on the last step

        cplx out[K1][8];
#pragma unroll
        for (int c = 0; c < K1; c++)
#pragma unroll
            for (int q = 0; q < 8; q++) out[c][q] = mk(0.0, 0.0);

        int32_t temp[16];
#pragma unroll 1
        for (int f = 0; f < F; f++) {
            const int c = f / L, p = f % L;        // component, digit index (0-based)
            if (p == 0) {
                int32_t cur[16];
#pragma unroll
                for (int m = 0; m < 16; m++) cur[m] = acc_lds[c * kN + lane + 64 * m];
                int a_here = a;
                asm volatile("" : "+v"(a_here));   // keeps the 32 rotate addresses/signs from being hoisted out of the f loop
                rotate_sub2(lane, a_here, acc_lds + c * kN, cur, P.g.offset, xormask, temp);
            }
            cplx x[8];
            load_digits2(temp, p + 1, beta, x);
            // pass A
            dft8<false>(x);
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
            x2_load(lane, x, xch);
            WAVE_LDS_FENCE();
            dft8<false>(x);
            // MAC: out[co] += D[p, c] .* BK_i[p, c].a[co]        (tgsw.jl:128)
            if (KPF == 16) {
#pragma unroll
                for (int co = 0; co < K1; co++)
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[co][k2] = cfma(x[k2], kbuf[co * 8 + k2], out[co][k2]);
            } else {
                const cplx *kp = key_ptr(i, f);
                cplx k1v[8];
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) k1v[k2] = kp[(8 + k2) * 64];
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) out[0][k2] = cfma(x[k2], kbuf[k2], out[0][k2]);
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) out[1][k2] = cfma(x[k2], k1v[k2], out[1][k2]);
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
            int32_t accr[16];
#pragma unroll
            for (int m = 0; m < 16; m++) accr[m] = acc_lds[co * kN + lane + 64 * m];
            untwist_add2(out[co], accr);
            store_acc<K1>(lane, accr, acc_lds + co * kN);
        }
        WAVE_LDS_FENCE();
    }

    int32_t *ext = P.ext + w * (kN + 1);
#pragma unroll
    for (int m = 0; m < 16; m++) {
        const int j = lane + 64 * m;
        const int32_t v = acc_lds[j];
        if (j == 0) ext[0] = v;
        else ext[kN - j] = (int32_t)(0u - (uint32_t)v);
    }
    if (lane == 0) ext[kN] = acc_lds[kN];
}

// ---- multi-key blind rotation (2 parties) ----------------------------------------------------------
// mk_internals.jl:464-495 (mk_mux_rotate, mk_blind_rotate, extract) with mk_tgsw_extern_mul (:348-391).
// Accumulator = P mask polynomials + body (P = 2): 3 polynomials in LDS.  Per step (party i, bit j):
// 3*L forward transforms, MAC against the expanded key polys x, y, c0, c1 of (i, j), 3 inverse
// transforms.  The reference inverse-transforms every product separately and sums in Int32
// (:359-366); summing in the spectrum domain first gives the same words (both are the exact product
// mod 2^32; rounding margin checked by the oracle test).
struct MkBrArgs {
    const int32_t *bara;  // [R][P*n+1]
    const cplx *bk;       // [P][n][2*L*P + 2*L][8][64] spectra (engine order, scaled 1/M)
    int32_t *ext;         // [R][P*N+1]
    Tables T;
    Gadget g;
    int32_t n;
    int32_t mu;
};

__device__ __forceinline__ void fft_fwd_wave(int lane, cplx (&x)[8], const cplx (&tw1f)[8], const cplx *tw2_lds, cplx *xch)
{
    dft8<false>(x);
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

template <int L, int PARTY>
__device__ __forceinline__ void mk_party_steps(int lane, const MkBrArgs &P, const int32_t *bara, int32_t *acc_lds,
                                               cplx *xch, const cplx *tw2_lds, const cplx (&tw1f)[8], int32_t xormask)
{
    constexpr int NP = 2;                         // parties
    constexpr int PER = 2 * L * NP + 2 * L;       // key polys per (party, bit)
    const int beta = P.g.log2_base;
#pragma unroll 1
    for (int j = 0; j < P.n; j++) {
        const int a = bara[PARTY * P.n + j] & (2 * kN - 1);
        const cplx *key = P.bk + ((size_t)PARTY * P.n + j) * PER * kM + lane;
        cplx out[NP + 1][8];
#pragma unroll
        for (int d = 0; d <= NP; d++)
#pragma unroll
            for (int q = 0; q < 8; q++) out[d][q] = mk(0.0, 0.0);
#pragma unroll
        for (int s = 0; s <= NP; s++) {           // source polynomial: masks 0..NP-1, body NP
            int32_t temp[16];
            {
                int32_t cur[16];
#pragma unroll
                for (int m = 0; m < 16; m++) cur[m] = acc_lds[s * kN + lane + 64 * m];
                int a_here = a;
                asm volatile("" : "+v"(a_here));
                rotate_sub2(lane, a_here, acc_lds + s * kN, cur, P.g.offset, xormask, temp);
            }
#pragma unroll 1
            for (int p = 0; p < L; p++) {
                cplx x[8];
                load_digits2(temp, p + 1, beta, x);
                fft_fwd_wave(lane, x, tw1f, tw2_lds, xch);
                // key polys for this transform (mk_internals.jl:371-385)
                const cplx *k_party, *k_body, *k_other = nullptr;
                if (s < NP) {
                    k_party = key + (size_t)(L * NP + p * NP + s) * kM;         // y[p, s]      -> a'_party
                    k_body = key + (size_t)(p * NP + s) * kM;                   // x[p, s]      -> b'
                    if (s != PARTY) k_other = key + (size_t)(L * NP + p * NP + PARTY) * kM;   // y[p, party] -> a'_s
                } else {
                    k_party = key + (size_t)(2 * L * NP + L + p) * kM;          // c1[p]        -> a'_party
                    k_body = key + (size_t)(2 * L * NP + p) * kM;               // c0[p]        -> b'
                }
                cplx kv[8];
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) kv[k2] = k_party[k2 * 64];
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) out[PARTY][k2] = cfma(x[k2], kv[k2], out[PARTY][k2]);
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) kv[k2] = k_body[k2 * 64];
#pragma unroll
                for (int k2 = 0; k2 < 8; k2++) out[NP][k2] = cfma(x[k2], kv[k2], out[NP][k2]);
                if (s < NP && s != PARTY) {
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) kv[k2] = k_other[k2 * 64];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[s < NP ? s : 0][k2] = cfma(x[k2], kv[k2], out[s < NP ? s : 0][k2]);
                }
            }
        }
#pragma unroll
        for (int d = 0; d <= NP; d++) {
            fft_inv_wave(lane, out[d], tw1f, tw2_lds, xch);
            int32_t accr[16];
#pragma unroll
            for (int m = 0; m < 16; m++) accr[m] = acc_lds[d * kN + lane + 64 * m];
            untwist_add2(out[d], accr);
            store_acc<2>(lane, accr, acc_lds + d * kN);
        }
        WAVE_LDS_FENCE();
    }
}

template <int L>
__global__ __launch_bounds__(64, 2) void mk_blind_rotate_kernel(MkBrArgs P)
{
    constexpr int NP = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem);                    // [NP+1][N]
    cplx *xch = reinterpret_cast<cplx *>(smem + (NP + 1) * kN * 4);          // [kXchElems]
    cplx *tw2_lds = xch + kXchElems;                                         // [8][8]
    const int lane = threadIdx.x;
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (NP * P.n + 1);
    const int32_t xormask = gadget_xor_mask(L, P.g.log2_base);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    tw2_lds[lane] = P.T.tw2[lane];
    {   // acc = (0, ..., 0, X^{-barb} * mu)       mk_internals.jl:491-492, 72-79
        const int barb = bara[NP * P.n] & (2 * kN - 1);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int idx = (lane + 64 * m + barb) & (2 * kN - 1);
            acc_lds[lane + 64 * m] = 0;
            acc_lds[kN + lane + 64 * m] = 0;
            acc_lds[2 * kN + lane + 64 * m] = (idx & kN) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
    }
    WAVE_LDS_FENCE();
    // party-major double loop (mk_internals.jl:475-476)
    mk_party_steps<L, 0>(lane, P, bara, acc_lds, xch, tw2_lds, tw1f, xormask);
    mk_party_steps<L, 1>(lane, P, bara, acc_lds, xch, tw2_lds, tw1f, xormask);

    // mk_tlwe_extract_sample (mk_internals.jl:88-95): one extracted mask column per party, b = body[0]
    int32_t *ext = P.ext + w * (NP * kN + 1);
#pragma unroll
    for (int c = 0; c < NP; c++)
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int jj = lane + 64 * m;
            const int32_t v = acc_lds[c * kN + jj];
            if (jj == 0) ext[c * kN] = v;
            else ext[c * kN + kN - jj] = (int32_t)(0u - (uint32_t)v);
        }
    if (lane == 0) ext[NP * kN] = acc_lds[NP * kN];
}

// ---- small batches: two waves per blind rotation ----------------------------------------------------
// With fewer rotations than wave slots (single gates, sequential circuits, small batches) one wave per
// rotation leaves the chip idle and a gate takes n x (4 forward + 2 inverse transforms) of latency.
// Here wave c (c = 0: mask polynomial, c = 1: body) owns accumulator polynomial c: it rotates and
// decomposes only its own polynomial, runs its L forward transforms, MACs both output components, hands
// the partial sum for the other component over through LDS (double-buffered, ONE barrier per step), adds
// what it receives, inverse-transforms its own component and updates its own polynomial.  Same arithmetic
// per rotation as blind_rotate_kernel_v3, about half the latency.
template <int L>
__global__ __launch_bounds__(128, 1) void blind_rotate_kernel_w2(BrArgs P)
{
    constexpr int K1 = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_all = reinterpret_cast<int32_t *>(smem);                        // [K1][N]
    cplx *xch_all = reinterpret_cast<cplx *>(smem + K1 * kN * 4);                // [2 waves][kXchElems]
    cplx *xfer = xch_all + 2 * kXchElems;                                        // [2 parity][2 waves][512]
    cplx *tw2_lds = xfer + 2 * 2 * kM;                                           // [8][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = tid >> 6;                                                     // wave = owned polynomial
    int32_t *acc_lds = acc_all + wv * kN;
    cplx *xch = xch_all + wv * kXchElems;
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.T.tw1f[q * 64 + lane];
    if (tid < 64) tw2_lds[tid] = P.T.tw2[tid];
    {
        const int barb = bara[P.n] & (2 * kN - 1);
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int idx = (lane + 64 * m + barb) & (2 * kN - 1);
            const int32_t body = (idx & kN) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
            acc_lds[lane + 64 * m] = wv ? body : 0;
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        const int a = bara[i] & (2 * kN - 1);
        // key polys of transform (p, c = wv): [i][p][c][co][8][64]
        const cplx *key = P.bk + (size_t)i * (L * K1 * K1 * kM) + (size_t)wv * K1 * kM + lane;
        cplx own[8], oth[8];
#pragma unroll
        for (int q = 0; q < 8; q++) { own[q] = mk(0.0, 0.0); oth[q] = mk(0.0, 0.0); }
        int32_t temp[16];
        {
            int32_t cur[16];
#pragma unroll
            for (int m = 0; m < 16; m++) cur[m] = acc_lds[lane + 64 * m];
            rotate_sub2(lane, a, acc_lds, cur, P.g.offset, xormask, temp);
        }
#pragma unroll 1
        for (int p = 0; p < L; p++) {
            cplx x[8];
            load_digits2(temp, p + 1, beta, x);
            const cplx *kp = key + (size_t)p * K1 * K1 * kM;
            cplx kown[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) kown[k2] = kp[(size_t)wv * kM + k2 * 64];           // co = wv (issued before the FFT)
            fft_fwd_wave(lane, x, tw1f, tw2_lds, xch);
            cplx koth[8];
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) koth[k2] = kp[(size_t)(1 - wv) * kM + k2 * 64];     // co = 1 - wv
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) own[k2] = cfma(x[k2], kown[k2], own[k2]);
#pragma unroll
            for (int k2 = 0; k2 < 8; k2++) oth[k2] = cfma(x[k2], koth[k2], oth[k2]);
        }
        // hand the other component's partial sum over (buffer by step parity: one barrier per step)
        cplx *mine = xfer + ((i & 1) * 2 + wv) * kM, *theirs = xfer + ((i & 1) * 2 + (1 - wv)) * kM;
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) mine[k2 * 64 + lane] = oth[k2];
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 8; k2++) own[k2] = cadd(own[k2], theirs[k2 * 64 + lane]);
        fft_inv_wave(lane, own, tw1f, tw2_lds, xch);
        int32_t accr[16];
#pragma unroll
        for (int m = 0; m < 16; m++) accr[m] = acc_lds[lane + 64 * m];
        untwist_add2(own, accr);
        store_acc<2>(lane, accr, acc_lds);
        WAVE_LDS_FENCE();
    }
    __syncthreads();
    int32_t *ext = P.ext + w * (kN + 1);
    if (wv == 0) {
#pragma unroll
        for (int m = 0; m < 16; m++) {
            const int j = lane + 64 * m;
            const int32_t v = acc_all[j];
            if (j == 0) ext[0] = v;
            else ext[kN - j] = (int32_t)(0u - (uint32_t)v);
        }
    } else if (lane == 0) {
        ext[kN] = acc_all[kN];
    }
}

// ---- blind rotation for tlwe_mask_size k = 2 (api.jl:30,55 keyword) ---------------------------------
// Same algorithm as blind_rotate_kernel_v3 with a 3-polynomial accumulator: 3*L forward transforms and
// 3 inverse transforms per step, out[co] += D[p, c] .* BK_i[p, c].a[co] for c, co in 0..2 (tgsw.jl:125-129).
template <int L>
__global__ __launch_bounds__(64, 2) void blind_rotate_kernel_k2(BrArgs P)
{
    constexpr int K1 = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem);                    // [K1][N]
    cplx *xch = reinterpret_cast<cplx *>(smem + K1 * kN * 4);
    cplx *tw2_lds = xch + kXchElems;
    const int lane = threadIdx.x;
    const size_t w = blockIdx.x;
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

#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        const int a = bara[i] & (2 * kN - 1);
        const cplx *key = P.bk + (size_t)i * (L * K1 * K1 * kM) + lane;
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
                cplx x[8];
                load_digits2(temp, p + 1, beta, x);
                fft_fwd_wave(lane, x, tw1f, tw2_lds, xch);
                const cplx *kp = key + (size_t)(p * K1 + c) * K1 * kM;
#pragma unroll
                for (int co = 0; co < K1; co++) {
                    cplx kv[8];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) kv[k2] = kp[(co * 8 + k2) * 64];
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
            untwist_add2(out[d], accr);
            store_acc<2>(lane, accr, acc_lds + d * kN);
        }
        WAVE_LDS_FENCE();
    }
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

// ---- N = 2048: two waves per blind rotation ----------------------------------------------------------
// M = 1024 folded points.  One radix-2 DIF stage is split across the two waves of a 128-thread block:
//   a_j = z_j + z_{j+512}  -> wave 0 -> even frequencies,   b_j = (z_j - z_{j+512}) W_1024^j -> wave 1 -> odd,
// then each wave runs the same 512-point transform as the N = 1024 kernels on its half, MACs its own
// frequencies, inverse-transforms them, and the halves are recombined through LDS (2 barriers per inverse
// transform).  With z_j = u_j w^j, w = e^{-i pi/2048}, w^512 = kappa = e^{-i pi/4}, j = t + 64 r:
//   wave 0 pass-A input  x_r = e^{-i pi r/32}  (u + kappa u'),   lane factor w^t             in tw1f
//   wave 1 pass-A input  x_r = e^{-i pi 5r/32} (u - kappa u'),   lane factor w^t W_1024^t    in tw1f
// Every wave rotates/decomposes all four coefficient classes it needs (t+64m, m < 32) itself.
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
    const int32_t *bara;   // [R][n+1]
    const cplx *bk;        // [n][L][2][2][2 (wave)][8][64]
    int32_t *ext;          // [R][N+1]
    const cplx *tw1f2;     // [2 (wave)][8][64]
    const cplx *tw2;       // [8][8]
    Gadget g;
    int32_t n, mu;
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

template <int MM>
__device__ __forceinline__ void rotate_sub_2048(int lane, int a, const int32_t *acc_lds, int32_t offset, int32_t xormask, int32_t (&temp)[32])
{
    const int base = (lane - a) & (2 * kN2 - 1);
#pragma unroll
    for (int m = 0; m < 32; m++) {
        const int idx = (base + 64 * m) & (2 * kN2 - 1);
        const int32_t v = acc_lds[idx & (kN2 - 1)];
        const int32_t cur = acc_lds[lane + 64 * m];
        const uint32_t sgn = (idx & kN2) ? 0xFFFFFFFFu : 0u;
        temp[m] = (int32_t)(((((uint32_t)v ^ sgn) - sgn) - (uint32_t)cur + (uint32_t)offset) ^ (uint32_t)xormask);
    }
}

// forward 512-point transform of this wave's half (after the radix-2 split), x in / spectrum out
__device__ __forceinline__ void fft_fwd_half(int lane, cplx (&x)[8], const cplx (&tw1f)[8], const cplx *tw2_lds, cplx *xch)
{
    fft_fwd_wave(lane, x, tw1f, tw2_lds, xch);
}

template <int L>
__global__ __launch_bounds__(128, 2) void blind_rotate_kernel_n2048(Br2048Args P)
{
    constexpr int K1 = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t *acc_lds = reinterpret_cast<int32_t *>(smem);                         // [K1][2048]
    cplx *xch_all = reinterpret_cast<cplx *>(smem + K1 * kN2 * 4);                // [2 waves][kXchElems]
    cplx *tw2_lds = xch_all + 2 * kXchElems;                                      // [8][8]
    const int tid = threadIdx.x, lane = tid & 63;
    const bool wave1 = (tid >> 6) != 0;                                           // wave-uniform
    cplx *xch = xch_all + (wave1 ? kXchElems : 0);
    cplx *xch_other = xch_all + (wave1 ? 0 : kXchElems);
    const size_t w = blockIdx.x;
    const int32_t *bara = P.bara + w * (P.n + 1);
    const int beta = P.g.log2_base;
    const int32_t xormask = gadget_xor_mask(L, beta);
    const double sg = wave1 ? -0.70710678118654752440 : 0.70710678118654752440;

    cplx tw1f[8];
#pragma unroll
    for (int q = 0; q < 8; q++) tw1f[q] = P.tw1f2[(wave1 ? 512 : 0) + q * 64 + lane];
    if (tid < 64) tw2_lds[tid] = P.tw2[tid];
    {
        const int barb = bara[P.n] & (2 * kN2 - 1);
        for (int j = tid; j < kN2; j += 128) {
            const int idx = (j + barb) & (2 * kN2 - 1);
            acc_lds[j] = 0;
            acc_lds[kN2 + j] = (idx & kN2) ? (int32_t)(0u - (uint32_t)P.mu) : P.mu;
        }
    }
    __syncthreads();

#pragma unroll 1
    for (int i = 0; i < P.n; i++) {
        const int a = bara[i] & (2 * kN2 - 1);
        const cplx *key = P.bk + (size_t)i * (L * K1 * K1 * 2 * kM) + (wave1 ? kM : 0) + lane;
        cplx out[K1][8];
#pragma unroll
        for (int d = 0; d < K1; d++)
#pragma unroll
            for (int q = 0; q < 8; q++) out[d][q] = mk(0.0, 0.0);
#pragma unroll 1
        for (int c = 0; c < K1; c++) {
            int32_t temp[32];
            {
                int a_here = a;
                asm volatile("" : "+v"(a_here));
                rotate_sub_2048<0>(lane, a_here, acc_lds + c * kN2, P.g.offset, xormask, temp);
            }
#pragma unroll 1
            for (int p = 0; p < L; p++) {
                cplx x[8];
#define FWD_IN(R)                                                                                          \
    {                                                                                                      \
        const int32_t lo = digit2(temp[R], p + 1, beta), l2 = digit2(temp[R + 8], p + 1, beta);            \
        const int32_t hi = digit2(temp[R + 16], p + 1, beta), h2 = digit2(temp[R + 24], p + 1, beta);      \
        x[R] = fwd_in_2048<R>((double)lo, (double)hi, (double)(l2 - h2), (double)(l2 + h2), sg, wave1);    \
    }
                FWD_IN(0) FWD_IN(1) FWD_IN(2) FWD_IN(3) FWD_IN(4) FWD_IN(5) FWD_IN(6) FWD_IN(7)
#undef FWD_IN
                fft_fwd_half(lane, x, tw1f, tw2_lds, xch);
                const cplx *kp = key + (size_t)(p * K1 + c) * K1 * 2 * kM;
#pragma unroll
                for (int co = 0; co < K1; co++) {
                    cplx kv[8];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) kv[k2] = kp[(size_t)co * 2 * kM + k2 * 64];
#pragma unroll
                    for (int k2 = 0; k2 < 8; k2++) out[co][k2] = cfma(x[k2], kv[k2], out[co][k2]);
                }
            }
        }
        __syncthreads();   // every rotated read of this step is done before anybody updates acc_lds
#pragma unroll
        for (int d = 0; d < K1; d++) {
            fft_inv_wave(lane, out[d], tw1f, tw2_lds, xch);          // alpha (wave 0) / beta (wave 1)
#pragma unroll
            for (int r = 0; r < 8; r++) xch[r * 64 + lane] = out[d][r];
            __syncthreads();
            cplx oth[8];
#pragma unroll
            for (int r = 0; r < 8; r++) oth[r] = xch_other[r * 64 + lane];
            __syncthreads();
            // wave 0: (conj(alpha) + conj(beta) e_r) c_r      -> coefficients jj, jj+1024
            // wave 1: (conj(alpha) - conj(beta) e_r) c_{r+8}  -> coefficients jj+512, jj+1536
#define COMBINE(R)                                                                                         \
    {                                                                                                      \
        const cplx al = wave1 ? oth[R] : out[d][R], be = wave1 ? out[d][R] : oth[R];                       \
        const double er = cos_pi32(4 * R), ei = -sin_pi32(4 * R);           /* e_r = e^{-i pi r/8} */      \
        const double br = be.x * er + be.y * ei, bi = be.x * ei - be.y * er; /* conj(beta) e_r: re, im */  \
        const double vr = wave1 ? al.x - br : al.x + br;                     /* conj(alpha) = (al.x, -al.y) */ \
        const double vi = wave1 ? -al.y - bi : -al.y + bi;                                                 \
        const double cr = wave1 ? cos_pi32(R + 8) : cos_pi32(R), ci = wave1 ? -sin_pi32(R + 8) : -sin_pi32(R); \
        const double re = vr * cr - vi * ci, im = vr * ci + vi * cr;                                       \
        const int jlo = lane + 64 * R + (wave1 ? 512 : 0);                                                 \
        int32_t *ap = acc_lds + d * kN2;                                                                   \
        ap[jlo] = (int32_t)((uint32_t)ap[jlo] + (uint32_t)round_to_torus32(re));                           \
        ap[jlo + 1024] = (int32_t)((uint32_t)ap[jlo + 1024] + (uint32_t)round_to_torus32(im));             \
    }
            COMBINE(0) COMBINE(1) COMBINE(2) COMBINE(3) COMBINE(4) COMBINE(5) COMBINE(6) COMBINE(7)
#undef COMBINE
        }
        __syncthreads();
    }

    int32_t *ext = P.ext + w * (kN2 + 1);
    for (int j = tid; j < kN2; j += 128) {
        const int32_t v = acc_lds[j];
        if (j == 0) ext[0] = v;
        else ext[kN2 - j] = (int32_t)(0u - (uint32_t)v);
    }
    if (tid == 0) ext[kN2] = acc_lds[kN2];
}

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
#define FWD_IN(R)                                                                                          \
    {                                                                                                      \
        const double lo = (double)poly[lane + 64 * R], l2 = (double)poly[lane + 64 * R + 512];             \
        const double hi = (double)poly[lane + 64 * R + 1024], h2 = (double)poly[lane + 64 * R + 1536];     \
        x[R] = fwd_in_2048<R>(lo, hi, l2 - h2, l2 + h2, sg, wave1);                                        \
    }
    FWD_IN(0) FWD_IN(1) FWD_IN(2) FWD_IN(3) FWD_IN(4) FWD_IN(5) FWD_IN(6) FWD_IN(7)
#undef FWD_IN
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

// Bootstrapping-key preparation: Int32 polynomial -> spectrum in the engine's order, scaled 1/M.
// (the analogue of forward_transform.(bk), bootstrap.jl:12)
__global__ __launch_bounds__(64) void bk_prepare_kernel(const int32_t *__restrict__ bk_i32, cplx *__restrict__ out, Tables T)
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
    const double s = 1.0 / kM;
#pragma unroll
    for (int k2 = 0; k2 < 8; k2++) out[q * kM + k2 * 64 + lane] = mk(x[k2].x * s, x[k2].y * s);
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

// keyswitch.jl:45-80.  One workgroup per output sample; thread w owns words w, w + blockDim, ...
// Input sample = ext[e0] (+ ext[e1] + (0, 2^29) for MUX, gates.jl:174).
struct KsArgs {
    const int32_t *ext;     // [R][kN+1]
    const int32_t *ks;      // [kN][t][base-1][n+1]
    const int32_t *e0;      // [G] index into ext
    const int32_t *e1;      // [G] second index or -1
    const int32_t *dst;     // [G] output gate index (NULL: identity)
    int32_t *out;           // [B][n+1]
    int32_t n, kN, t, log2_base;
};

template <int WPT>  // words per thread
__global__ __launch_bounds__(256) void keyswitch_kernel(KsArgs P)
{
    const int g = blockIdx.x;
    const int tid = threadIdx.x;
    const int n1 = P.n + 1;
    const int32_t *x0 = P.ext + (size_t)P.e0[g] * (P.kN + 1);
    const int e1 = P.e1 ? P.e1[g] : -1;
    const int32_t *x1 = e1 >= 0 ? P.ext + (size_t)e1 * (P.kN + 1) : nullptr;
    const int base1 = (1 << P.log2_base) - 1;
    const uint32_t prec_offset = 1u << (32 - (1 + P.log2_base * P.t));       // keyswitch.jl:58

    uint32_t accw[WPT];
#pragma unroll
    for (int u = 0; u < WPT; u++) accw[u] = 0;

    for (int i = 0; i < P.kN; i++) {
        uint32_t ai = (uint32_t)x0[i];
        if (x1) ai += (uint32_t)x1[i];
        const int32_t aibar = (int32_t)(ai + prec_offset);                  // keyswitch.jl:59
        const int32_t *rows_i = P.ks + (size_t)i * P.t * base1 * n1;
        for (int j = 1; j <= P.t; j++) {
            const int d = (aibar >> (32 - j * P.log2_base)) & base1;         // keyswitch.jl:65-67
            if (d != 0) {                                                    // keyswitch.jl:73
                const int32_t *row = rows_i + (size_t)((j - 1) * base1 + (d - 1)) * n1;
#pragma unroll
                for (int u = 0; u < WPT; u++) {
                    const int wd = tid + u * 256;
                    if (wd < n1) accw[u] -= (uint32_t)row[wd];               // keyswitch.jl:74
                }
            }
        }
    }
    const size_t og = P.dst ? (size_t)P.dst[g] : (size_t)g;
    int32_t *o = P.out + og * n1;
#pragma unroll
    for (int u = 0; u < WPT; u++) {
        const int wd = tid + u * 256;
        if (wd < n1) {
            uint32_t v = accw[u];
            if (wd == P.n) {                                                 // keyswitch.jl:50
                v += (uint32_t)x0[P.kN];
                if (x1) v += (uint32_t)x1[P.kN] + (1u << 29);                // gates.jl:174
            }
            o[wd] = (int32_t)v;
        }
    }
}

// keyswitch v2: a workgroup owns G output samples and every output word (thread t -> word t), so each
// keyswitch-key row is fetched once per G samples instead of once per sample (the v1 kernel moves
// B * 6144 rows * 2 KB through L2; this one B/G * 24576 rows).  The digits of a sample are wave-uniform:
// the rounded mask words sit in LDS, are read as a broadcast and moved to SGPRs with readfirstlane, and
// the row to subtract is selected with scalar masks (no divergent branches, no row for digit 0 -
// keyswitch.jl:32-38,73-75).
template <int G, int IB /* mask words staged in LDS per block */>
__global__ __launch_bounds__(512) void keyswitch_kernel_v2(KsArgs P, int B)
{
    __shared__ int32_t abar[G][IB];
    const int tid = threadIdx.x;
    const int n1 = P.n + 1;
    const int g0 = blockIdx.x * G;
    const int t_len = P.t, lb = P.log2_base;
    const int base1 = (1 << lb) - 1;
    const uint32_t prec_offset = 1u << (32 - (1 + lb * t_len));             // keyswitch.jl:58
    const bool active = tid < n1;
    const int wd = active ? tid : 0;

    uint32_t acc[G];
#pragma unroll
    for (int g = 0; g < G; g++) acc[g] = 0;

    for (int i0 = 0; i0 < P.kN; i0 += IB) {
        __syncthreads();
        // stage aibar = a_i + prec_offset for G samples x IB mask words (MUX: sum of two extracted samples)
        for (int idx = tid; idx < G * IB; idx += 512) {
            const int g = idx / IB, ii = idx % IB;
            const int gg = min(g0 + g, B - 1);
            const int i = i0 + ii;
            uint32_t ai = 0;
            if (i < P.kN) {
                ai = (uint32_t)P.ext[(size_t)P.e0[gg] * (P.kN + 1) + i];
                const int e1 = P.e1 ? P.e1[gg] : -1;
                if (e1 >= 0) ai += (uint32_t)P.ext[(size_t)e1 * (P.kN + 1) + i];
                ai += prec_offset;                                           // keyswitch.jl:59
            } else {
                ai = 0;   // digits all zero: contributes nothing
            }
            abar[g][ii] = (int32_t)ai;
        }
        __syncthreads();
        const int iend = min(IB, P.kN - i0);
        for (int ii = 0; ii < iend; ii++) {
            const int32_t *rows_i = P.ks + (size_t)(i0 + ii) * t_len * base1 * n1 + wd;
            for (int j = 0; j < t_len; j++) {
                // the (base-1) candidate rows for digit position j; base = 4 in every shipped set,
                // general base handled by the loop over h
                const int32_t *rj = rows_i + (size_t)j * base1 * n1;
                const int sh = 32 - (j + 1) * lb;
                if (base1 == 3) {
                    const uint32_t r1 = (uint32_t)rj[0], r2 = (uint32_t)rj[n1], r3 = (uint32_t)rj[2 * n1];
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        const int a = __builtin_amdgcn_readfirstlane(abar[g][ii]);
                        const int d = (a >> sh) & 3;                         // keyswitch.jl:65-67
                        const uint32_t m1 = d == 1 ? 0xFFFFFFFFu : 0u, m2 = d == 2 ? 0xFFFFFFFFu : 0u,
                                       m3 = d == 3 ? 0xFFFFFFFFu : 0u;
                        acc[g] -= (r1 & m1) | (r2 & m2) | (r3 & m3);         // keyswitch.jl:73-75
                    }
                } else {
                    for (int h = 1; h <= base1; h++) {
                        const uint32_t r = (uint32_t)rj[(size_t)(h - 1) * n1];
#pragma unroll
                        for (int g = 0; g < G; g++) {
                            const int a = __builtin_amdgcn_readfirstlane(abar[g][ii]);
                            const int d = (a >> sh) & base1;
                            acc[g] -= (d == h) ? r : 0u;
                        }
                    }
                }
            }
        }
    }
    if (!active) return;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int gg = g0 + g;
        if (gg >= B) break;
        uint32_t v = acc[g];
        if (tid == P.n) {                                                    // keyswitch.jl:50
            v += (uint32_t)P.ext[(size_t)P.e0[gg] * (P.kN + 1) + P.kN];
            const int e1 = P.e1 ? P.e1[gg] : -1;
            if (e1 >= 0) v += (uint32_t)P.ext[(size_t)e1 * (P.kN + 1) + P.kN] + (1u << 29);   // gates.jl:174
        }
        const size_t og = P.dst ? (size_t)P.dst[gg] : (size_t)gg;
        P.out[og * n1 + tid] = (int32_t)v;
    }
}

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

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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

__global__ __launch_bounds__(256) void keyswitch_kernel_v4(Ks4Args P)
{
    constexpr int MT = 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int wt = blockIdx.y;
    const int gbase = (blockIdx.x * 4 + wave) * (32 * MT);
    if (gbase >= P.G) return;

    const int32_t *row0[MT], *row1[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int gg = min(gbase + mt * 32 + c, P.G - 1);
        row0[mt] = P.ext + (size_t)P.e0[gg] * P.in_stride + P.in_off;
        const int e1 = P.e1 ? P.e1[gg] : -1;
        row1[mt] = e1 >= 0 ? P.ext + (size_t)e1 * P.in_stride + P.in_off : nullptr;
    }
    const uint32_t prec_offset = 1u << 15;                                   // 2^(32 - (1 + 2*8))   keyswitch.jl:58

    i32x16 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int pl = 0; pl < 4; pl++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[mt][pl][r] = 0;

    const i32x4 *bp = P.bmat + (size_t)wt * 4 * 64 + lane;
    const size_t bstep = (size_t)P.wtiles * 4 * 64;                          // fragments per mask word i
    i32x4 bcur[4], bnxt[4];
#pragma unroll
    for (int pl = 0; pl < 4; pl++) bcur[pl] = bp[pl * 64];

    const int i_begin = (int)blockIdx.z * (P.kN / P.kslices), i_end = i_begin + P.kN / P.kslices;
    bp += (size_t)i_begin * bstep;
#pragma unroll
    for (int pl = 0; pl < 4; pl++) bcur[pl] = bp[pl * 64];
    for (int i4 = i_begin; i4 < i_end; i4 += 4) {
        // 4 consecutive mask words of this lane's samples (MUX: sum of two extracted samples, gates.jl:174)
        uint32_t a4[MT][4];
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int ii = 0; ii < 4; ii++) {
                uint32_t v = (uint32_t)row0[mt][i4 + ii];
                if (row1[mt]) v += (uint32_t)row1[mt][i4 + ii];
                a4[mt][ii] = v + prec_offset;                                // keyswitch.jl:59
            }
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            const int i = i4 + ii;
            const i32x4 *bn = bp + (size_t)((i + 1 < i_end ? i + 1 : i) - i_begin) * bstep;
#pragma unroll
            for (int pl = 0; pl < 4; pl++) bnxt[pl] = bn[pl * 64];
            i32x4 afrag[MT];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    // digit position j = 4h+q (0-based) occupies bits [30-2j, 31-2j]   keyswitch.jl:65-67
                    const uint32_t d8 = (a4[mt][ii] >> (27 - 2 * (4 * h + q))) & 24u;   // 8 * digit
                    afrag[mt][q] = (int32_t)(1u << d8);                      // one-hot byte
                }
#pragma unroll
            for (int pl = 0; pl < 4; pl++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt][pl] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afrag[mt], bcur[pl], acc[mt][pl], 0, 0, 0);
#pragma unroll
            for (int pl = 0; pl < 4; pl++) bcur[pl] = bnxt[pl];
        }
    }

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

// gate_not / gate_constant / copy (gates.jl:76-93)
__global__ void trivial_gates_kernel(const int32_t *in0, const int32_t *__restrict__ src_rows,
                                     const int32_t *__restrict__ dst_rows, const uint8_t *__restrict__ ops,
                                     int32_t *out, int n)
{
    const size_t gs = (size_t)src_rows[blockIdx.x], gd = (size_t)dst_rows[blockIdx.x];
    const int op = ops[blockIdx.x];
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        uint32_t v;
        if (op == TFHE_GATE_NOT) v = 0u - (uint32_t)in0[gs * (n + 1) + i];
        else if (op == TFHE_GATE_COPY) v = (uint32_t)in0[gs * (n + 1) + i];
        else v = (i == n) ? (op == TFHE_GATE_CONST1 ? (1u << 29) : 0u - (1u << 29)) : 0u;
        out[gd * (n + 1) + i] = (int32_t)v;
    }
}

// extracted sample copy-out for tfhe_bootstrap_batch(with_keyswitch = 0) is a plain memcpy.

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct tfhe_ctx {
    tfhe_params P{};
    int device = 0;
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // batch start, BR start/end(=KS start), KS end
    bool timing_valid = false;
    int64_t last_rotations = 0;
    int ks_variant = 4;          // 1 = one workgroup per sample, 2 = gate-tiled, 3 = tiled + sliced + XCD-aware, 4 = int8 MFMA (default)
    int64_t br_small = 512;      // batches of at most this many rotations use the two-waves-per-rotation kernel (-1: never)
    int br_variant = 2;          // 1 = baseline kernel, 2 = v3 full-chunk key prefetch (default), 3 = v3 half-chunk

    // tables
    cplx *d_tables = nullptr;   // tw1[512] | tw2[64] | twist[512]
    Tables T{};
    Gadget g{};

    // keys
    cplx *d_bk = nullptr;       size_t bk_polys = 0;
    int32_t *d_ks = nullptr;
    int32_t *d_ksp = nullptr;   int ks_stride = 0;   // row-padded copy for keyswitch_kernel_v3
    void *d_ks4 = nullptr;      int ks4_wtiles = 0;  // MFMA B fragments for keyswitch_kernel_v4 (base 4, t = 8)
    void *d_mk_ks4 = nullptr;   size_t mk_ks4_frags = 0;
    bool have_bk = false, have_ks = false;
    // multi-key (2 parties)
    cplx *d_mk_bk = nullptr;
    int32_t *d_mk_ksp = nullptr;   // [P] row-padded keyswitch keys back to back
    size_t mk_ksp_words = 0;       // words per party in d_mk_ksp
    int mk_parties = 0;
    bool have_mk_bk = false, have_mk_ks = false;

    // device-resident wire table for levelised circuits: int32 [num_wires][n+1]
    int32_t *d_wires = nullptr; int64_t num_wires = 0;

    // workspaces
    DevBuf bara, ext, map, io[4];
    void *h_map = nullptr; size_t h_map_cap = 0;   // pinned staging for the index maps
    hipEvent_t map_ev = nullptr; bool map_pending = false;   // guards reuse of h_map

    int set_err(int code, const char *fmt, ...)
    {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        err = buf;
        return code;
    }
};

#define HIP_TRY(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return (ctx)->set_err(TFHE_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static void build_tables(std::vector<cplx> &h)
{
    h.resize(kTableElems + 1024);
    fill_tables<long double>(h.data(), [](long double a) { return cosl(a); }, [](long double a) { return sinl(a); });
    // N = 2048: tw1f2[w][q][t] = e^{-i pi t (1 + 4w + 8q) / 2048}
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int w = 0; w < 2; w++)
        for (int q = 0; q < 8; q++)
            for (int t = 0; t < 64; t++) {
                const long double a = -pi * (long double)(t * (1 + 4 * w + 8 * q)) / 2048.0L;
                h[kTableElems + w * 512 + q * 64 + t] = mk((double)cosl(a), (double)sinl(a));
            }
}

static int ilog2i(int x) { int r = 0; while ((1 << r) < x) r++; return r; }

extern "C" {

int32_t tfhe_abi_version(void) { return TFHE_MI355X_ABI_VERSION; }

int32_t tfhe_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

const char *tfhe_last_error(const tfhe_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t tfhe_ctx_create(const tfhe_params *params, int32_t device_id, tfhe_ctx **out_ctx)
{
    if (!params || !out_ctx) { g_create_error = "tfhe_ctx_create: NULL argument"; return TFHE_ERR_INVALID_ARG; }
    *out_ctx = nullptr;
    const tfhe_params &p = *params;
    char buf[256];
    auto fail = [&](int code, const char *msg) { g_create_error = msg; return code; };
    if (p.n < 1 || p.N < 2 || (p.N & (p.N - 1)) || p.k < 1 || p.bs_l < 1 || p.bs_log2_base < 1 || p.ks_t < 1 ||
        p.ks_log2_base < 1 || p.parties < 1)
        return fail(TFHE_ERR_INVALID_ARG, "tfhe_ctx_create: parameters must be positive and N a power of two");
    if (p.bs_l * p.bs_log2_base > 32) return fail(TFHE_ERR_INVALID_ARG, "tfhe_ctx_create: bs_l * bs_log2_base > 32");
    if (p.ks_t * p.ks_log2_base > 31) return fail(TFHE_ERR_INVALID_ARG, "tfhe_ctx_create: ks_t * ks_log2_base > 31");
    if (p.N != kN && p.N != kN2) {
        snprintf(buf, sizeof buf, "tfhe_ctx_create: this build supports N = %d or %d (got %d)", kN, kN2, p.N);
        return fail(TFHE_ERR_UNSUPPORTED, buf);
    }
    if (p.N == kN2 && (p.k != 1 || p.parties != 1))
        return fail(TFHE_ERR_UNSUPPORTED, "tfhe_ctx_create: N = 2048 is supported with tlwe_mask_size 1, single key");
    if (p.k != 1 && p.k != 2) return fail(TFHE_ERR_UNSUPPORTED, "tfhe_ctx_create: this build supports tlwe_mask_size k = 1 or 2");
    if (p.k != 1 && p.parties != 1) return fail(TFHE_ERR_UNSUPPORTED, "tfhe_ctx_create: multi-key needs tlwe_mask_size 1 (as the reference, mk_internals.jl:89-91)");
    if (p.bs_l > 4) return fail(TFHE_ERR_UNSUPPORTED, "tfhe_ctx_create: bs_decomp_length > 4 unsupported");
    if (p.n + 1 > 1024) return fail(TFHE_ERR_UNSUPPORTED, "tfhe_ctx_create: lwe_size + 1 > 1024 unsupported");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(buf, sizeof buf, "tfhe_ctx_create: no HIP device available (%s)", hipGetErrorString(e));
        return fail(TFHE_ERR_DEVICE, buf);
    }
    if (device_id < 0 || device_id >= ndev) return fail(TFHE_ERR_INVALID_ARG, "tfhe_ctx_create: device_id out of range");

    tfhe_ctx *c = new tfhe_ctx();
    c->P = p;
    c->device = device_id;
    c->g = make_gadget(p.bs_l, p.bs_log2_base);
    auto bail = [&](hipError_t err, const char *what) {
        snprintf(buf, sizeof buf, "tfhe_ctx_create: %s failed: %s", what, hipGetErrorString(err));
        g_create_error = buf;
        tfhe_ctx_destroy(c);
        return (int32_t)TFHE_ERR_DEVICE;
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    for (auto &ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->map_ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    std::vector<cplx> h;
    build_tables(h);
    if ((e = hipMalloc((void **)&c->d_tables, h.size() * sizeof(cplx))) != hipSuccess) return bail(e, "hipMalloc(tables)");
    if ((e = hipMemcpy(c->d_tables, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice)) != hipSuccess)
        return bail(e, "hipMemcpy(tables)");
    c->T = tables_from(c->d_tables);
    *out_ctx = c;
    return TFHE_OK;
}

void tfhe_ctx_destroy(tfhe_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_bk) (void)hipFree(c->d_bk);
    if (c->d_ks) (void)hipFree(c->d_ks);
    if (c->d_ksp) (void)hipFree(c->d_ksp);
    if (c->d_ks4) (void)hipFree(c->d_ks4);
    if (c->d_wires) (void)hipFree(c->d_wires);
    if (c->d_mk_ks4) (void)hipFree(c->d_mk_ks4);
    if (c->d_mk_bk) (void)hipFree(c->d_mk_bk);
    if (c->d_mk_ksp) (void)hipFree(c->d_mk_ksp);
    c->bara.release(); c->ext.release(); c->map.release();
    for (auto &b : c->io) b.release();
    if (c->h_map) (void)hipHostFree(c->h_map);
    for (auto &ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    if (c->map_ev) (void)hipEventDestroy(c->map_ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int32_t tfhe_ctx_params(const tfhe_ctx *ctx, tfhe_params *out)
{
    if (!ctx || !out) return TFHE_ERR_INVALID_ARG;
    *out = ctx->P;
    return TFHE_OK;
}

static size_t bk_poly_count(const tfhe_params &p) { return (size_t)p.n * p.bs_l * (p.k + 1) * (p.k + 1); }

static int32_t load_bk_common(tfhe_ctx *c, const void *host, size_t bytes_in, bool is_c128)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!host) return c->set_err(TFHE_ERR_INVALID_ARG, "load_bootstrap_key: NULL key pointer");
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "load_bootstrap_key: context is multi-key, use tfhe_mk_load_*");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t npolys = bk_poly_count(c->P);
    const bool big = (c->P.N == kN2);
    if (c->d_bk) { (void)hipFree(c->d_bk); c->d_bk = nullptr; c->have_bk = false; }
    HIP_TRY(c, hipMalloc((void **)&c->d_bk, npolys * (size_t)(c->P.N / 2) * sizeof(cplx)));
    void *d_in = nullptr;
    HIP_TRY(c, hipMalloc(&d_in, bytes_in));
    hipError_t e = hipMemcpyAsync(d_in, host, bytes_in, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        if (big && is_c128)
            hipLaunchKernelGGL(bk_permute_c128_kernel_n2048, dim3((unsigned)npolys), dim3(128), 0, c->stream, (const cplx *)d_in, c->d_bk);
        else if (big)
            hipLaunchKernelGGL(bk_prepare_kernel_n2048, dim3((unsigned)npolys), dim3(128), 0, c->stream, (const int32_t *)d_in, c->d_bk,
                               (const cplx *)(c->d_tables + kTableElems), c->T.tw2);
        else if (is_c128)
            hipLaunchKernelGGL(bk_permute_c128_kernel, dim3((unsigned)npolys), dim3(64), 0, c->stream, (const cplx *)d_in, c->d_bk);
        else
            hipLaunchKernelGGL(bk_prepare_kernel, dim3((unsigned)npolys), dim3(64), 0, c->stream, (const int32_t *)d_in, c->d_bk, c->T);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_in);
    if (e != hipSuccess) return c->set_err(TFHE_ERR_DEVICE, "load_bootstrap_key: %s", hipGetErrorString(e));
    c->bk_polys = npolys;
    c->have_bk = true;
    return TFHE_OK;
}

int32_t tfhe_load_bootstrap_key_i32(tfhe_ctx *c, const int32_t *bk)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    return load_bk_common(c, bk, bk_poly_count(c->P) * (size_t)c->P.N * sizeof(int32_t), false);
}

int32_t tfhe_load_bootstrap_key_c128(tfhe_ctx *c, const double *bk_spectra)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    return load_bk_common(c, bk_spectra, bk_poly_count(c->P) * (size_t)(c->P.N / 2) * sizeof(cplx), true);
}

static size_t ks_word_count(const tfhe_params &p)
{
    return (size_t)p.k * p.N * p.ks_t * ((1u << p.ks_log2_base) - 1) * (size_t)(p.n + 1);
}

int32_t tfhe_load_keyswitch_key(tfhe_ctx *c, const int32_t *ks)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!ks) return c->set_err(TFHE_ERR_INVALID_ARG, "load_keyswitch_key: NULL key pointer");
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "load_keyswitch_key: context is multi-key, use tfhe_mk_load_*");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = ks_word_count(c->P) * sizeof(int32_t);
    if (c->d_ks) { (void)hipFree(c->d_ks); c->d_ks = nullptr; c->have_ks = false; }
    HIP_TRY(c, hipMalloc((void **)&c->d_ks, bytes));
    HIP_TRY(c, hipMemcpy(c->d_ks, ks, bytes, hipMemcpyHostToDevice));
    {   // row-padded copy: stride = n+1 rounded up to 4 words so that rows are 16-byte aligned
        const size_t n1 = (size_t)c->P.n + 1, stride = (n1 + 3) & ~(size_t)3;
        const size_t rows = ks_word_count(c->P) / n1;
        if (c->d_ksp) { (void)hipFree(c->d_ksp); c->d_ksp = nullptr; }
        HIP_TRY(c, hipMalloc((void **)&c->d_ksp, rows * stride * 4));
        HIP_TRY(c, hipMemset(c->d_ksp, 0, rows * stride * 4));
        HIP_TRY(c, hipMemcpy2D(c->d_ksp, stride * 4, c->d_ks, n1 * 4, n1 * 4, rows, hipMemcpyDeviceToDevice));
        c->ks_stride = (int)stride;
    }
    if (c->d_ks4) { (void)hipFree(c->d_ks4); c->d_ks4 = nullptr; }
    if (c->P.ks_log2_base == 2 && c->P.ks_t == 8 && (c->P.k * c->P.N) % 4 == 0) {
        const int kNn = c->P.k * c->P.N, wtiles = (c->P.n + 1 + 31) / 32;
        const size_t frags = (size_t)kNn * wtiles * 4 * 64;
        HIP_TRY(c, hipMalloc(&c->d_ks4, frags * 16));
        hipLaunchKernelGGL(ks4_prepare_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, c->stream, (const int32_t *)c->d_ks,
                           (i32x4 *)c->d_ks4, c->P.n, kNn, wtiles);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->ks4_wtiles = wtiles;
    }
    c->have_ks = true;
    return TFHE_OK;
}

// ---- launch helpers ------------------------------------------------------------------------------
static int32_t launch_blind_rotate(tfhe_ctx *c, size_t R, int32_t mu, hipStream_t s)
{
    BrArgs a;
    a.bara = (const int32_t *)c->bara.p;
    a.bk = c->d_bk;
    a.ext = (int32_t *)c->ext.p;
    a.T = c->T;
    a.g = c->g;
    a.n = c->P.n;
    a.mu = mu;
    if (c->P.N == kN2) {
        Br2048Args b;
        b.bara = a.bara; b.bk = a.bk; b.ext = a.ext; b.tw1f2 = c->d_tables + kTableElems; b.tw2 = c->T.tw2; b.g = c->g; b.n = a.n; b.mu = mu;
        const size_t ldsb = 2 * kN2 * 4 + (2 * kXchElems + 64) * sizeof(cplx);
        switch (c->P.bs_l) {
        case 1: hipLaunchKernelGGL((blind_rotate_kernel_n2048<1>), dim3((unsigned)R), dim3(128), ldsb, s, b); break;
        case 2: hipLaunchKernelGGL((blind_rotate_kernel_n2048<2>), dim3((unsigned)R), dim3(128), ldsb, s, b); break;
        case 3: hipLaunchKernelGGL((blind_rotate_kernel_n2048<3>), dim3((unsigned)R), dim3(128), ldsb, s, b); break;
        case 4: hipLaunchKernelGGL((blind_rotate_kernel_n2048<4>), dim3((unsigned)R), dim3(128), ldsb, s, b); break;
        default: return c->set_err(TFHE_ERR_UNSUPPORTED, "blind rotate: bs_l = %d unsupported", c->P.bs_l);
        }
        HIP_TRY(c, hipGetLastError());
        return TFHE_OK;
    }
    if (c->P.k == 2) {
        const size_t ldsk = 3 * kN * 4 + (kXchElems + 64) * sizeof(cplx);
        switch (c->P.bs_l) {
        case 1: hipLaunchKernelGGL((blind_rotate_kernel_k2<1>), dim3((unsigned)R), dim3(64), ldsk, s, a); break;
        case 2: hipLaunchKernelGGL((blind_rotate_kernel_k2<2>), dim3((unsigned)R), dim3(64), ldsk, s, a); break;
        case 3: hipLaunchKernelGGL((blind_rotate_kernel_k2<3>), dim3((unsigned)R), dim3(64), ldsk, s, a); break;
        case 4: hipLaunchKernelGGL((blind_rotate_kernel_k2<4>), dim3((unsigned)R), dim3(64), ldsk, s, a); break;
        default: return c->set_err(TFHE_ERR_UNSUPPORTED, "blind rotate: bs_l = %d unsupported", c->P.bs_l);
        }
        HIP_TRY(c, hipGetLastError());
        return TFHE_OK;
    }
    if ((c->br_small >= 0 && (int64_t)R <= c->br_small) && c->br_variant >= 2) {
        const size_t ldsw = 2 * kN * 4 + (2 * kXchElems + 4 * kM + 64) * sizeof(cplx);
        switch (c->P.bs_l) {
        case 1: hipLaunchKernelGGL((blind_rotate_kernel_w2<1>), dim3((unsigned)R), dim3(128), ldsw, s, a); break;
        case 2: hipLaunchKernelGGL((blind_rotate_kernel_w2<2>), dim3((unsigned)R), dim3(128), ldsw, s, a); break;
        case 3: hipLaunchKernelGGL((blind_rotate_kernel_w2<3>), dim3((unsigned)R), dim3(128), ldsw, s, a); break;
        case 4: hipLaunchKernelGGL((blind_rotate_kernel_w2<4>), dim3((unsigned)R), dim3(128), ldsw, s, a); break;
        default: return c->set_err(TFHE_ERR_UNSUPPORTED, "blind rotate: bs_l = %d unsupported", c->P.bs_l);
        }
        HIP_TRY(c, hipGetLastError());
        return TFHE_OK;
    }
    if (c->br_variant >= 2) {
        const size_t lds3 = 2 * kN * 4 + (kXchElems + 64) * sizeof(cplx);
#define LAUNCH_V3(LL, KK) hipLaunchKernelGGL((blind_rotate_kernel_v3<LL, KK, false>), dim3((unsigned)R), dim3(64), lds3, s, a)
        const bool half = (c->br_variant == 3);
        switch (c->P.bs_l) {
        case 1: if (half) LAUNCH_V3(1, 8); else LAUNCH_V3(1, 16); break;
        case 2: if (half) LAUNCH_V3(2, 8); else LAUNCH_V3(2, 16); break;
        case 3: if (half) LAUNCH_V3(3, 8); else LAUNCH_V3(3, 16); break;
        case 4: if (half) LAUNCH_V3(4, 8); else LAUNCH_V3(4, 16); break;
        default: return c->set_err(TFHE_ERR_UNSUPPORTED, "blind rotate: bs_l = %d unsupported", c->P.bs_l);
        }
#undef LAUNCH_V3
        HIP_TRY(c, hipGetLastError());
        return TFHE_OK;
    }
    const size_t lds = 2 * kN * 4 + kXchElems * sizeof(cplx);
    switch (c->P.bs_l) {
    case 1: hipLaunchKernelGGL((blind_rotate_kernel<1, 2>), dim3((unsigned)R), dim3(64), lds, s, a); break;
    case 2: hipLaunchKernelGGL((blind_rotate_kernel<2, 2>), dim3((unsigned)R), dim3(64), lds, s, a); break;
    case 3: hipLaunchKernelGGL((blind_rotate_kernel<3, 2>), dim3((unsigned)R), dim3(64), lds, s, a); break;
    case 4: hipLaunchKernelGGL((blind_rotate_kernel<4, 2>), dim3((unsigned)R), dim3(64), lds, s, a); break;
    default: return c->set_err(TFHE_ERR_UNSUPPORTED, "blind rotate: bs_l = %d unsupported", c->P.bs_l);
    }
    HIP_TRY(c, hipGetLastError());
    return TFHE_OK;
}

static int32_t launch_keyswitch(tfhe_ctx *c, size_t G, const int32_t *e0, const int32_t *e1, const int32_t *dst,
                                const int32_t *ext, int32_t *out, hipStream_t s)
{
    KsArgs k;
    k.ext = ext;
    k.ks = c->d_ks;
    k.e0 = e0; k.e1 = e1; k.dst = dst;
    k.out = out;
    k.n = c->P.n; k.kN = c->P.k * c->P.N; k.t = c->P.ks_t; k.log2_base = c->P.ks_log2_base;
    const int n1 = c->P.n + 1;
    if (c->ks_variant == 4 && c->d_ks4) {
        Ks4Args a4;
        a4.ext = ext; a4.bmat = (const i32x4 *)c->d_ks4; a4.e0 = e0; a4.e1 = e1; a4.dst = dst; a4.out = out;
        a4.n = c->P.n; a4.kN = k.kN; a4.G = (int)G; a4.wtiles = c->ks4_wtiles;
        a4.in_stride = k.kN + 1; a4.in_off = 0; a4.in_b = k.kN; a4.out_stride = n1; a4.out_off = 0; a4.out_b = c->P.n; a4.add_b = 1;
        a4.kslices = (G <= 512 && k.kN % 64 == 0) ? 16 : 1;                 // small batches: split the mask words over 16 blocks
        if (a4.kslices > 1) {
            Ks3Args i3;
            i3.ext = ext; i3.e0 = e0; i3.e1 = e1; i3.dst = dst; i3.out = out; i3.kN = k.kN; i3.n = c->P.n;
            i3.in_stride = a4.in_stride; i3.in_b = a4.in_b; i3.out_stride = n1; i3.out_b = c->P.n;
            hipLaunchKernelGGL(ks3_init_kernel, dim3((unsigned)G), dim3(256), 0, s, i3);
        }
        hipLaunchKernelGGL(keyswitch_kernel_v4, dim3((unsigned)((G + 255) / 256), (unsigned)c->ks4_wtiles, (unsigned)a4.kslices), dim3(256), 0, s, a4);
        HIP_TRY(c, hipGetLastError());
        return TFHE_OK;
    }
    if ((c->ks_variant == 3 || c->ks_variant == 4) && c->P.ks_log2_base == 2 && c->P.ks_t % 4 == 0 && k.kN % (KS3_SLICES * 1) == 0 &&
        k.kN / KS3_SLICES <= 128) {
        Ks3Args a3;
        a3.ext = ext; a3.ksp = c->d_ksp; a3.e0 = e0; a3.e1 = e1; a3.dst = dst; a3.out = out;
        a3.n = c->P.n; a3.kN = k.kN; a3.t = c->P.ks_t; a3.log2_base = 2; a3.stride = c->ks_stride; a3.G = (int)G;
        a3.in_stride = k.kN + 1; a3.in_off = 0; a3.in_b = k.kN; a3.out_stride = n1; a3.out_off = 0; a3.out_b = c->P.n;
        hipLaunchKernelGGL(ks3_init_kernel, dim3((unsigned)G), dim3(256), 0, s, a3);
        const unsigned tiles = (unsigned)((G + KS3_G - 1) / KS3_G);
        hipLaunchKernelGGL(keyswitch_kernel_v3, dim3(tiles * KS3_SLICES, (unsigned)((c->ks_stride + 511) / 512)), dim3(128), 0, s, a3);
        HIP_TRY(c, hipGetLastError());
        return TFHE_OK;
    }
    if (c->ks_variant == 2 && n1 <= 512) {
        constexpr int KG = 16;
        hipLaunchKernelGGL((keyswitch_kernel_v2<KG, 128>), dim3((unsigned)((G + KG - 1) / KG)), dim3(512), 0, s, k, (int)G);
        HIP_TRY(c, hipGetLastError());
        return TFHE_OK;
    }
    if (n1 <= 256) hipLaunchKernelGGL((keyswitch_kernel<1>), dim3((unsigned)G), dim3(256), 0, s, k);
    else if (n1 <= 512) hipLaunchKernelGGL((keyswitch_kernel<2>), dim3((unsigned)G), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((keyswitch_kernel<4>), dim3((unsigned)G), dim3(256), 0, s, k);
    HIP_TRY(c, hipGetLastError());
    return TFHE_OK;
}

static int32_t ensure_host_map(tfhe_ctx *c, size_t bytes)
{
    if (c->map_pending) {   // the previous call's H2D copy of the staging block must have been consumed
        HIP_TRY(c, hipEventSynchronize(c->map_ev));
        c->map_pending = false;
    }
    if (bytes <= c->h_map_cap) return TFHE_OK;
    if (c->h_map) (void)hipHostFree(c->h_map);
    c->h_map = nullptr; c->h_map_cap = 0;
    HIP_TRY(c, hipHostMalloc(&c->h_map, bytes + bytes / 4 + 256, hipHostMallocDefault));
    c->h_map_cap = bytes + bytes / 4 + 256;
    return TFHE_OK;
}

// Common body of tfhe_gates_batch_dev (operands = rows g of three arrays, ia = ib = ic = io = NULL) and
// tfhe_gates_level (operands = rows ia[g], ib[g], ic[g] of one wire table, result row io[g]).
static int32_t run_gates(tfhe_ctx *c, const char *who, const uint8_t *opcodes, int64_t B, const int32_t *d_in0,
                         const int32_t *d_in1, const int32_t *d_in2, int32_t *d_out, const int32_t *ia, const int32_t *ib,
                         const int32_t *ic, const int32_t *io, hipStream_t s)
{
    // classify gates: rotations (R), keyswitches (G), trivial (T)
    size_t R = 0, G = 0, Tn = 0;
    bool need1 = false, need2 = false, need0 = false;
    for (int64_t g = 0; g < B; g++) {
        const int op = opcodes[g];
        if (op >= TFHE_GATE__COUNT) return c->set_err(TFHE_ERR_INVALID_ARG, "%s: bad opcode %d at gate %lld", who, op, (long long)g);
        if (op == TFHE_GATE_MUX) { R += 2; G += 1; need0 = need1 = need2 = true; }
        else if (op == TFHE_GATE_NOT || op == TFHE_GATE_COPY) { Tn++; need0 = true; }
        else if (op == TFHE_GATE_CONST0 || op == TFHE_GATE_CONST1) { Tn++; }
        else { R += 1; G += 1; need0 = need1 = true; }
    }
    if ((need0 && !d_in0) || (need1 && !d_in1) || (need2 && !d_in2))
        return c->set_err(TFHE_ERR_INVALID_ARG, "%s: an operand array required by the opcodes is NULL", who);
    if (R > 0 && (!c->have_bk || !c->have_ks)) return c->set_err(TFHE_ERR_NO_KEY, "%s: bootstrapping/keyswitch key not loaded", who);

    // index maps, one pinned staging block:
    //   rot_a[R] | rot_b[R] | ks_e0[G] | ks_e1[G] | ks_dst[G] | triv_src[T] | triv_dst[T] | rot_kind[R] | triv_op[T]
    const size_t map_bytes = (2 * R + 3 * G + 2 * Tn) * 4 + R + Tn;
    int32_t rc = ensure_host_map(c, map_bytes);
    if (rc) return rc;
    int32_t *h_ra = (int32_t *)c->h_map, *h_rb = h_ra + R;
    int32_t *h_e0 = h_rb + R, *h_e1 = h_e0 + G, *h_dst = h_e1 + G, *h_ts = h_dst + G, *h_td = h_ts + Tn;
    uint8_t *h_kind = (uint8_t *)(h_td + Tn), *h_top = h_kind + R;
    {
        size_t r = 0, k = 0, t = 0;
        for (int64_t g = 0; g < B; g++) {
            const int op = opcodes[g];
            const int32_t ra = ia ? ia[g] : (int32_t)g, rb = ib ? ib[g] : (int32_t)g, rcw = ic ? ic[g] : (int32_t)g;
            const int32_t ro = io ? io[g] : (int32_t)g;
            if (op == TFHE_GATE_MUX) {
                h_ra[r] = ra; h_rb[r] = rb; h_kind[r] = 100;           // AND(x, y)      gates.jl:166
                h_ra[r + 1] = ra; h_rb[r + 1] = rcw; h_kind[r + 1] = 101;   // AND(NOT x, z)  gates.jl:170
                h_e0[k] = (int32_t)r; h_e1[k] = (int32_t)(r + 1); h_dst[k] = ro;
                r += 2; k++;
            } else if (op == TFHE_GATE_NOT || op == TFHE_GATE_COPY || op == TFHE_GATE_CONST0 || op == TFHE_GATE_CONST1) {
                h_ts[t] = ra; h_td[t] = ro; h_top[t] = (uint8_t)op; t++;
            } else {
                h_ra[r] = ra; h_rb[r] = rb; h_kind[r] = (uint8_t)op;
                h_e0[k] = (int32_t)r; h_e1[k] = -1; h_dst[k] = ro;
                r++; k++;
            }
        }
    }
    HIP_TRY(c, c->map.reserve(map_bytes));
    HIP_TRY(c, hipMemcpyAsync(c->map.p, c->h_map, map_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipEventRecord(c->map_ev, s));
    c->map_pending = true;
    const int32_t *d_ra = (const int32_t *)c->map.p, *d_rb = d_ra + R;
    const int32_t *d_e0 = d_rb + R, *d_e1 = d_e0 + G, *d_dst = d_e1 + G, *d_ts = d_dst + G, *d_td = d_ts + Tn;
    const uint8_t *d_kind = (const uint8_t *)(d_td + Tn), *d_top = d_kind + R;

    const int n = c->P.n, kNn = c->P.k * c->P.N;
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    if (R > 0) {
        HIP_TRY(c, c->bara.reserve(R * (size_t)(n + 1) * 4));
        HIP_TRY(c, c->ext.reserve(R * (size_t)(kNn + 1) * 4));
        hipLaunchKernelGGL(prologue_kernel, dim3((unsigned)R), dim3(256), 0, s, d_in0, d_in1, d_in2, d_ra, d_rb, d_kind,
                           (int32_t *)c->bara.p, n, ilog2i(2 * c->P.N));
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    if (R > 0) {
        rc = launch_blind_rotate(c, R, (int32_t)(1u << 29), s);   // mu = encode_message(1, 8), gates.jl:17
        if (rc) return rc;
    }
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    if (G > 0) {
        rc = launch_keyswitch(c, G, d_e0, d_e1, d_dst, (const int32_t *)c->ext.p, d_out, s);
        if (rc) return rc;
    }
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    if (Tn > 0) {
        hipLaunchKernelGGL(trivial_gates_kernel, dim3((unsigned)Tn), dim3(256), 0, s, d_in0, d_ts, d_td, d_top, d_out, n);
        HIP_TRY(c, hipGetLastError());
    }
    c->timing_valid = true;
    c->last_rotations = (int64_t)R;
    return TFHE_OK;
}

int32_t tfhe_gates_batch_dev(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *d_in0, const int32_t *d_in1,
                             const int32_t *d_in2, int32_t *d_out, int64_t B, void *stream)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!opcodes || !d_out))) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: NULL argument or negative B");
    if (B == 0) { c->timing_valid = false; c->last_rotations = 0; return TFHE_OK; }
    if (B > (int64_t)1 << 30) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: B too large");
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "gates_batch: context is multi-key");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return run_gates(c, "gates_batch", opcodes, B, d_in0, d_in1, d_in2, d_out, nullptr, nullptr, nullptr, nullptr, s);
}

// ---- levelised circuit execution on a device-resident wire table -------------------------
int32_t tfhe_wires_alloc(tfhe_ctx *c, int64_t num_wires)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (num_wires < 0 || num_wires > ((int64_t)1 << 30)) return c->set_err(TFHE_ERR_INVALID_ARG, "wires_alloc: bad wire count");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_wires) { (void)hipFree(c->d_wires); c->d_wires = nullptr; c->num_wires = 0; }
    if (num_wires == 0) return TFHE_OK;
    HIP_TRY(c, hipMalloc((void **)&c->d_wires, (size_t)num_wires * (c->P.n + 1) * 4));
    c->num_wires = num_wires;
    return TFHE_OK;
}

static int32_t wires_range_ok(tfhe_ctx *c, const char *who, int64_t first, int64_t count, const void *host)
{
    if (!c->d_wires) return c->set_err(TFHE_ERR_STATE, "%s: no wire table allocated", who);
    if (first < 0 || count < 0 || first + count > c->num_wires || (count > 0 && !host))
        return c->set_err(TFHE_ERR_INVALID_ARG, "%s: wire range [%lld, %lld) outside the table of %lld wires or NULL buffer", who,
                          (long long)first, (long long)(first + count), (long long)c->num_wires);
    return TFHE_OK;
}

int32_t tfhe_wires_upload(tfhe_ctx *c, int64_t first, int64_t count, const int32_t *host)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    int32_t rc = wires_range_ok(c, "wires_upload", first, count, host);
    if (rc || count == 0) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t row = (size_t)(c->P.n + 1) * 4;
    HIP_TRY(c, hipMemcpyAsync((char *)c->d_wires + (size_t)first * row, host, (size_t)count * row, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int32_t tfhe_wires_download(tfhe_ctx *c, int64_t first, int64_t count, int32_t *host)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    int32_t rc = wires_range_ok(c, "wires_download", first, count, host);
    if (rc || count == 0) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t row = (size_t)(c->P.n + 1) * 4;
    HIP_TRY(c, hipMemcpyAsync(host, (const char *)c->d_wires + (size_t)first * row, (size_t)count * row, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int32_t tfhe_gates_level(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *a, const int32_t *b, const int32_t *cc,
                         const int32_t *out, int64_t B)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!opcodes || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "gates_level: context is multi-key");
    if (!c->d_wires) return c->set_err(TFHE_ERR_STATE, "gates_level: no wire table allocated");
    // every index in range; no wire both written and read inside one level (the level's gates are independent)
    std::vector<uint8_t> mark((size_t)c->num_wires, 0);
    auto bad = [&](int64_t v) { return v < 0 || v >= c->num_wires; };
    for (int64_t g = 0; g < B; g++) {
        const int op = opcodes[g];
        if (op >= TFHE_GATE__COUNT) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: bad opcode %d at gate %lld", op, (long long)g);
        const bool has_a = !(op == TFHE_GATE_CONST0 || op == TFHE_GATE_CONST1);
        const bool has_b = has_a && !(op == TFHE_GATE_NOT || op == TFHE_GATE_COPY);
        const bool has_c = (op == TFHE_GATE_MUX);
        if ((has_a && (!a || bad(a[g]))) || (has_b && (!b || bad(b[g]))) || (has_c && (!cc || bad(cc[g]))) || bad(out[g]))
            return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: wire index out of range (or missing operand array) at gate %lld", (long long)g);
        if (mark[(size_t)out[g]] & 1) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: wire %d written twice in one level", out[g]);
        mark[(size_t)out[g]] |= 1;
    }
    for (int64_t g = 0; g < B; g++) {
        const int op = opcodes[g];
        const bool has_a = !(op == TFHE_GATE_CONST0 || op == TFHE_GATE_CONST1);
        const bool has_b = has_a && !(op == TFHE_GATE_NOT || op == TFHE_GATE_COPY);
        if ((has_a && (mark[(size_t)a[g]] & 1)) || (has_b && (mark[(size_t)b[g]] & 1)) || (op == TFHE_GATE_MUX && (mark[(size_t)cc[g]] & 1)))
            return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: gate %lld reads a wire written in the same level", (long long)g);
    }
    HIP_TRY(c, hipSetDevice(c->device));
    // operands of opcodes that ignore them get a valid dummy row (0)
    std::vector<int32_t> ia((size_t)B), ib((size_t)B), ic((size_t)B);
    for (int64_t g = 0; g < B; g++) {
        const int op = opcodes[g];
        const bool has_a = !(op == TFHE_GATE_CONST0 || op == TFHE_GATE_CONST1);
        const bool has_b = has_a && !(op == TFHE_GATE_NOT || op == TFHE_GATE_COPY);
        ia[(size_t)g] = has_a ? a[g] : 0;
        ib[(size_t)g] = has_b ? b[g] : 0;
        ic[(size_t)g] = op == TFHE_GATE_MUX ? cc[g] : 0;
    }
    return run_gates(c, "gates_level", opcodes, B, c->d_wires, c->d_wires, c->d_wires, c->d_wires, ia.data(), ib.data(), ic.data(), out,
                     c->stream);
}

int32_t tfhe_gates_batch(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1,
                         const int32_t *in2, int32_t *out, int64_t B)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!opcodes || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)B * (c->P.n + 1) * 4;
    const int32_t *hin[3] = {in0, in1, in2};
    int32_t *din[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; i++) {
        if (!hin[i]) continue;
        HIP_TRY(c, c->io[i].reserve(bytes));
        HIP_TRY(c, hipMemcpyAsync(c->io[i].p, hin[i], bytes, hipMemcpyHostToDevice, c->stream));
        din[i] = (int32_t *)c->io[i].p;
    }
    HIP_TRY(c, c->io[3].reserve(bytes));
    int32_t rc = tfhe_gates_batch_dev(c, opcodes, din[0], din[1], din[2], (int32_t *)c->io[3].p, B, nullptr);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

int32_t tfhe_bootstrap_batch(tfhe_ctx *c, int32_t mu, const int32_t *in, int32_t *out, int64_t B, int32_t with_keyswitch)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!in || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "bootstrap_batch: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "bootstrap_batch: context is multi-key");
    if (!c->have_bk || (with_keyswitch && !c->have_ks)) return c->set_err(TFHE_ERR_NO_KEY, "bootstrap_batch: key not loaded");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = c->P.n, kNn = c->P.k * c->P.N;
    const size_t in_bytes = (size_t)B * (n + 1) * 4;
    HIP_TRY(c, c->io[0].reserve(in_bytes));
    HIP_TRY(c, hipMemcpyAsync(c->io[0].p, in, in_bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(c, c->bara.reserve((size_t)B * (n + 1) * 4));
    HIP_TRY(c, c->ext.reserve((size_t)B * (kNn + 1) * 4));
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    hipLaunchKernelGGL(modswitch_kernel, dim3((unsigned)B), dim3(256), 0, s, (const int32_t *)c->io[0].p, (int32_t *)c->bara.p, n,
                       ilog2i(2 * c->P.N));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    int32_t rc = launch_blind_rotate(c, (size_t)B, mu, s);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    if (with_keyswitch) {
        // identity maps: e0[g] = g
        rc = ensure_host_map(c, (size_t)B * 4);
        if (rc) return rc;
        for (int64_t g = 0; g < B; g++) ((int32_t *)c->h_map)[g] = (int32_t)g;
        HIP_TRY(c, c->map.reserve((size_t)B * 4));
        HIP_TRY(c, hipMemcpyAsync(c->map.p, c->h_map, (size_t)B * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(c, c->io[3].reserve(in_bytes));
        rc = launch_keyswitch(c, (size_t)B, (const int32_t *)c->map.p, nullptr, nullptr, (const int32_t *)c->ext.p, (int32_t *)c->io[3].p, s);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(c->ev[3], s));
        HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, in_bytes, hipMemcpyDeviceToHost, s));
    } else {
        HIP_TRY(c, hipEventRecord(c->ev[3], s));
        HIP_TRY(c, hipMemcpyAsync(out, c->ext.p, (size_t)B * (kNn + 1) * 4, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    c->timing_valid = true;
    c->last_rotations = B;
    return TFHE_OK;
}

int32_t tfhe_keyswitch_batch(tfhe_ctx *c, const int32_t *in, int32_t *out, int64_t B)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!in || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "keyswitch_batch: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "keyswitch_batch: context is multi-key");
    if (!c->have_ks) return c->set_err(TFHE_ERR_NO_KEY, "keyswitch_batch: keyswitch key not loaded");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int n = c->P.n, kNn = c->P.k * c->P.N;
    const size_t in_bytes = (size_t)B * (kNn + 1) * 4, out_bytes = (size_t)B * (n + 1) * 4;
    HIP_TRY(c, c->ext.reserve(in_bytes));
    HIP_TRY(c, hipMemcpyAsync(c->ext.p, in, in_bytes, hipMemcpyHostToDevice, s));
    int32_t rc = ensure_host_map(c, (size_t)B * 4);
    if (rc) return rc;
    for (int64_t g = 0; g < B; g++) ((int32_t *)c->h_map)[g] = (int32_t)g;
    HIP_TRY(c, c->map.reserve((size_t)B * 4));
    HIP_TRY(c, hipMemcpyAsync(c->map.p, c->h_map, (size_t)B * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, c->io[3].reserve(out_bytes));
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    rc = launch_keyswitch(c, (size_t)B, (const int32_t *)c->map.p, nullptr, nullptr, (const int32_t *)c->ext.p, (int32_t *)c->io[3].p, s);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, out_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->timing_valid = true;
    c->last_rotations = 0;
    return TFHE_OK;
}

int32_t tfhe_mk_load_bootstrap_key_i32(tfhe_ctx *c, const int32_t *bk, int32_t parties)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!bk) return c->set_err(TFHE_ERR_INVALID_ARG, "mk_load_bootstrap_key: NULL key pointer");
    if (parties != 2 || c->P.parties < 2) return c->set_err(TFHE_ERR_UNSUPPORTED, "mk_load_bootstrap_key: this build supports exactly 2 parties on a context created with parties >= 2");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t per = (size_t)2 * c->P.bs_l * parties + 2 * c->P.bs_l;
    const size_t npolys = (size_t)parties * c->P.n * per;
    if (c->d_mk_bk) { (void)hipFree(c->d_mk_bk); c->d_mk_bk = nullptr; c->have_mk_bk = false; }
    HIP_TRY(c, hipMalloc((void **)&c->d_mk_bk, npolys * kM * sizeof(cplx)));
    void *d_in = nullptr;
    HIP_TRY(c, hipMalloc(&d_in, npolys * kN * 4));
    hipError_t e = hipMemcpyAsync(d_in, bk, npolys * kN * 4, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(bk_prepare_kernel, dim3((unsigned)npolys), dim3(64), 0, c->stream, (const int32_t *)d_in, c->d_mk_bk, c->T);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_in);
    if (e != hipSuccess) return c->set_err(TFHE_ERR_DEVICE, "mk_load_bootstrap_key: %s", hipGetErrorString(e));
    c->mk_parties = parties;
    c->have_mk_bk = true;
    return TFHE_OK;
}

int32_t tfhe_mk_load_keyswitch_key(tfhe_ctx *c, const int32_t *ks, int32_t parties)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!ks) return c->set_err(TFHE_ERR_INVALID_ARG, "mk_load_keyswitch_key: NULL key pointer");
    if (parties != 2 || c->P.parties < 2) return c->set_err(TFHE_ERR_UNSUPPORTED, "mk_load_keyswitch_key: this build supports exactly 2 parties on a context created with parties >= 2");
    if (c->P.ks_log2_base != 2 || c->P.ks_t % 4 != 0 || c->P.N % KS3_SLICES != 0 || c->P.N / KS3_SLICES > 128)
        return c->set_err(TFHE_ERR_UNSUPPORTED, "mk_load_keyswitch_key: keyswitch base must be 4 and t a multiple of 4");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n1 = (size_t)c->P.n + 1, stride = (n1 + 3) & ~(size_t)3;
    const size_t rows = (size_t)c->P.N * c->P.ks_t * ((1u << c->P.ks_log2_base) - 1);   // per party (k = 1)
    if (c->d_mk_ksp) { (void)hipFree(c->d_mk_ksp); c->d_mk_ksp = nullptr; c->have_mk_ks = false; }
    HIP_TRY(c, hipMalloc((void **)&c->d_mk_ksp, (size_t)parties * rows * stride * 4));
    HIP_TRY(c, hipMemset(c->d_mk_ksp, 0, (size_t)parties * rows * stride * 4));
    HIP_TRY(c, hipMemcpy2D(c->d_mk_ksp, stride * 4, ks, n1 * 4, n1 * 4, (size_t)parties * rows, hipMemcpyHostToDevice));
    c->mk_ksp_words = rows * stride;
    c->ks_stride = (int)stride;
    if (c->d_mk_ks4) { (void)hipFree(c->d_mk_ks4); c->d_mk_ks4 = nullptr; }
    if (c->P.ks_t == 8) {   // MFMA fragments per party (keyswitch_kernel_v4)
        const int wtiles = (c->P.n + 1 + 31) / 32;
        const size_t frags = (size_t)c->P.N * wtiles * 4 * 64, words = rows * n1;
        int32_t *d_tmp = nullptr;
        HIP_TRY(c, hipMalloc((void **)&d_tmp, words * 4));
        HIP_TRY(c, hipMalloc(&c->d_mk_ks4, (size_t)parties * frags * 16));
        for (int p = 0; p < parties; p++) {
            HIP_TRY(c, hipMemcpy(d_tmp, ks + (size_t)p * words, words * 4, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(ks4_prepare_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, c->stream, (const int32_t *)d_tmp,
                               (i32x4 *)c->d_mk_ks4 + (size_t)p * frags, c->P.n, c->P.N, wtiles);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        (void)hipFree(d_tmp);
        c->mk_ks4_frags = frags;
        c->ks4_wtiles = wtiles;
    }
    c->have_mk_ks = true;
    return TFHE_OK;
}

int32_t tfhe_mk_gate_nand_batch(tfhe_ctx *c, const int32_t *in0, const int32_t *in1, int32_t *out, int64_t B)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!in0 || !in1 || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "mk_gate_nand_batch: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (!c->have_mk_bk || !c->have_mk_ks) return c->set_err(TFHE_ERR_NO_KEY, "mk_gate_nand_batch: multi-key keys not loaded");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int NP = c->mk_parties, n = c->P.n, nw = NP * n + 1, ew = NP * kN + 1;
    const size_t bytes = (size_t)B * nw * 4;
    for (int i = 0; i < 2; i++) {
        HIP_TRY(c, c->io[i].reserve(bytes));
        HIP_TRY(c, hipMemcpyAsync(c->io[i].p, i == 0 ? in0 : in1, bytes, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(c, c->io[3].reserve(bytes));
    HIP_TRY(c, c->bara.reserve(bytes));
    HIP_TRY(c, c->ext.reserve((size_t)B * ew * 4));
    // maps: rot_gate[g] = g, kind = NAND, e0[g] = g
    int32_t rc = ensure_host_map(c, (size_t)B * 5);
    if (rc) return rc;
    int32_t *h_gate = (int32_t *)c->h_map;
    uint8_t *h_kind = (uint8_t *)(h_gate + B);
    for (int64_t g = 0; g < B; g++) { h_gate[g] = (int32_t)g; h_kind[g] = TFHE_GATE_NAND; }
    HIP_TRY(c, c->map.reserve((size_t)B * 5));
    HIP_TRY(c, hipMemcpyAsync(c->map.p, c->h_map, (size_t)B * 5, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipEventRecord(c->map_ev, s));
    c->map_pending = true;
    const int32_t *d_gate = (const int32_t *)c->map.p;
    const uint8_t *d_kind = (const uint8_t *)(d_gate + B);
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    // mk_gate_nand prologue (mk_gates.jl:8-10) = the NAND affine form over P*n+1 words, then mod-switch
    hipLaunchKernelGGL(prologue_kernel, dim3((unsigned)B), dim3(256), 0, s, (const int32_t *)c->io[0].p, (const int32_t *)c->io[1].p,
                       (const int32_t *)nullptr, d_gate, d_gate, d_kind, (int32_t *)c->bara.p, NP * n, ilog2i(2 * c->P.N));
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    MkBrArgs a;
    a.bara = (const int32_t *)c->bara.p; a.bk = c->d_mk_bk; a.ext = (int32_t *)c->ext.p; a.T = c->T; a.g = c->g;
    a.n = n; a.mu = (int32_t)(1u << 29);
    const size_t lds = 3 * kN * 4 + (kXchElems + 64) * sizeof(cplx);
    switch (c->P.bs_l) {
    case 2: hipLaunchKernelGGL((mk_blind_rotate_kernel<2>), dim3((unsigned)B), dim3(64), lds, s, a); break;
    case 3: hipLaunchKernelGGL((mk_blind_rotate_kernel<3>), dim3((unsigned)B), dim3(64), lds, s, a); break;
    case 4: hipLaunchKernelGGL((mk_blind_rotate_kernel<4>), dim3((unsigned)B), dim3(64), lds, s, a); break;
    default: return c->set_err(TFHE_ERR_UNSUPPORTED, "mk blind rotate: bs_l = %d unsupported", c->P.bs_l);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    // mk_keyswitch (mk_internals.jl:397-411): per party a single-key keyswitch of its mask column with b = 0
    if (c->ks_variant == 4 && c->d_mk_ks4) {
        Ks4Args a4;
        a4.ext = (const int32_t *)c->ext.p; a4.e0 = d_gate; a4.e1 = nullptr; a4.dst = nullptr; a4.out = (int32_t *)c->io[3].p;
        a4.n = n; a4.kN = kN; a4.G = (int)B; a4.wtiles = c->ks4_wtiles;
        a4.in_stride = ew; a4.in_b = NP * kN; a4.out_stride = nw; a4.out_b = NP * n;
        for (int p = 0; p < NP; p++) {
            a4.in_off = p * kN; a4.out_off = p * n; a4.add_b = (p == 0); a4.kslices = 1;
            a4.bmat = (const i32x4 *)c->d_mk_ks4 + (size_t)p * c->mk_ks4_frags;
            hipLaunchKernelGGL(keyswitch_kernel_v4, dim3((unsigned)((B + 255) / 256), (unsigned)c->ks4_wtiles), dim3(256), 0, s, a4);
        }
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev[3], s));
        HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, bytes, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        c->timing_valid = true;
        c->last_rotations = B;
        return TFHE_OK;
    }
    Ks3Args k3;
    k3.ext = (const int32_t *)c->ext.p; k3.e0 = d_gate; k3.e1 = nullptr; k3.dst = nullptr; k3.out = (int32_t *)c->io[3].p;
    k3.n = n; k3.kN = kN; k3.t = c->P.ks_t; k3.log2_base = 2; k3.stride = c->ks_stride; k3.G = (int)B;
    k3.in_stride = ew; k3.in_b = NP * kN; k3.out_stride = nw; k3.out_b = NP * n;
    k3.in_off = 0; k3.out_off = 0; k3.ksp = c->d_mk_ksp;
    hipLaunchKernelGGL(ks3_init_kernel, dim3((unsigned)B), dim3(256), 0, s, k3);
    const unsigned tiles = (unsigned)((B + KS3_G - 1) / KS3_G);
    for (int p = 0; p < NP; p++) {
        k3.in_off = p * kN; k3.out_off = p * n; k3.ksp = c->d_mk_ksp + (size_t)p * c->mk_ksp_words;
        hipLaunchKernelGGL(keyswitch_kernel_v3, dim3(tiles * KS3_SLICES, (unsigned)((c->ks_stride + 511) / 512)), dim3(128), 0, s, k3);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->timing_valid = true;
    c->last_rotations = B;
    return TFHE_OK;
}

int32_t tfhe_last_timing_ms(tfhe_ctx *c, int32_t which, float *ms)
{
    if (!c || !ms) return TFHE_ERR_INVALID_ARG;
    if (!c->timing_valid) return c->set_err(TFHE_ERR_STATE, "last_timing: no batch call recorded");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->ev[3]));
    int a, b;
    switch (which) {
    case 0: a = 1; b = 2; break;
    case 1: a = 2; b = 3; break;
    case 2: a = 0; b = 3; break;
    default: return c->set_err(TFHE_ERR_INVALID_ARG, "last_timing: which must be 0, 1 or 2");
    }
    HIP_TRY(c, hipEventElapsedTime(ms, c->ev[a], c->ev[b]));
    return TFHE_OK;
}

int64_t tfhe_last_rotation_count(const tfhe_ctx *c) { return c ? c->last_rotations : -1; }

int32_t tfhe_set_option(tfhe_ctx *c, const char *name, int64_t value)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!name || !*name) return TFHE_OK;
    if (!strcmp(name, "br_variant")) {
        if (value < 1 || value > 3) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: br_variant must be 1, 2 or 3");
        c->br_variant = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "br_small")) { c->br_small = value; return TFHE_OK; }
    if (!strcmp(name, "ks_variant")) {
        if (value < 1 || value > 4) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: ks_variant must be 1..4");
        c->ks_variant = (int)value;
        return TFHE_OK;
    }
    return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: unknown option '%s'", name);
}

}  // extern "C"
