// Device-side generation of the cloud key: BootstrapKey = tgsw_encrypt(s_i) for every bit of the LWE key
// (bootstrap.jl:6-15, tgsw.jl:52-88, tlwe.jl:63-73) and KeyswitchKey (keyswitch.jl:14-41, lwe.jl:49-55), in the
// canonical Int32 layouts of include/tfhe_mi355x.h.  One-off work (the reference does it on the host in make_key_pair,
// api.jl:139-146); here it saves the 82-620 MB host-to-device copy and the host's FFT products for large sets.
//
// Randomness: Philox4x32-10 (Salmon et al., SC'11), counter-based, so every word of the key is a pure function of
// (seed, position) and the test suite restates the streams in numpy (tests/philox_ref.py).  The reference's
// MersenneTwister stream is not reproduced (keys are data that crosses the boundary).  The caller's seed is SIX 32-bit
// words with two roles that must not share key material:
//   seed[0..1]  key of the MASK streams 1 and 3: their output is published as the `a` words of the cloud key anyway;
//   seed[2..5]  128 bits that are AS SECRET AS THE SECRET KEY: they key the NOISE streams 2 and 4 (Philox key =
//               seed[2..3], counter words 2 and 3 = stream ^ seed[4], seed[5]).  Whoever knows them can subtract every
//               noise term and solve b - e = <a, s> for the secret keys; and because they are independent of the mask
//               key, the public mask words give no handle for guessing them.  Philox is a statistical generator with no
//               PRF security claim: it is used here as a NON-CRYPTOGRAPHIC EXPANDER of that 128-bit secret, which the
//               caller must draw from a cryptographic source (the shipped wrappers take it from os.urandom /
//               RandomDevice() whatever generator they were handed; tests pass fixed words).
//   stream 1: bootstrap-key mask words      index = ((sample * k) + c) * N + coefficient
//   stream 2: bootstrap-key noise           index = sample * N + coefficient
//   stream 3: keyswitch-key mask words      index = sample * n + m
//   stream 4: keyswitch-key noise           index = sample
// Uniform word of index i = output (i & 3) of the block with counter (i >> 2); Gaussian of index i = Box-Muller on
// outputs 0, 1 of the block with counter i.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace keygen {

struct U4 { uint32_t x, y, z, w; };

__host__ __device__ inline U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

struct Seed { uint32_t w[6]; };      // [0..1] mask key (public), [2..5] noise key (secret)

__device__ inline uint32_t uniform_word(uint32_t stream, uint64_t index, const Seed &sd)
{
    const uint64_t blk = index >> 2;
    const U4 r = philox4x32_10((uint32_t)blk, (uint32_t)(blk >> 32), stream, 0u, sd.w[0], sd.w[1]);
    const uint32_t sel = (uint32_t)index & 3u;
    return sel == 0 ? r.x : sel == 1 ? r.y : sel == 2 ? r.z : r.w;
}

// standard normal: sqrt(-2 ln u1) cos(2 pi u2), u = (word + 0.5) / 2^32
__device__ inline double gaussian(uint32_t stream, uint64_t index, const Seed &sd)
{
    const U4 r = philox4x32_10((uint32_t)index, (uint32_t)(index >> 32), stream ^ sd.w[4], sd.w[5], sd.w[2], sd.w[3]);
    const double u1 = ((double)r.x + 0.5) * (1.0 / 4294967296.0), u2 = ((double)r.y + 0.5) * (1.0 / 4294967296.0);
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2);
}

// numeric-functions.jl:51-53: trunc(Int32, d * 2^32), wrapping
__device__ inline uint32_t dtot32(double d)
{
    return (uint32_t)(int32_t)(int64_t)trunc(d * 4294967296.0);
}

struct Args {
    const int32_t *lwe_key;    // [n] 0/1
    const int32_t *tlwe_key;   // [k][N] 0/1
    int32_t *bk;               // [n][l][k+1][k+1][N]
    int32_t *ks;               // [k*N][t][base-1][n+1]
    double *ks_noise;          // [k*N * t * (base-1)] scratch
    double *ks_mean;           // [1] scratch
    int32_t n, N, k, l, beta, t, ks_log2_base;
    double bs_alpha, ks_alpha;
    Seed seed;
};

// One workgroup per TLWE sample (i, p, j): k uniform mask polynomials, body = sum_c a_c * s_c + e (tlwe.jl:63-73), plus
// the gadget message s_i * 2^(32 - (p+1) beta) on coefficient 0 of component j (tgsw.jl:60-70).  The products by the
// binary key polynomials are exact: signed sums of rotations out of LDS.
__global__ __launch_bounds__(256) void bk_kernel(Args A)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint32_t *a_lds = reinterpret_cast<uint32_t *>(smem);                 // [k][N]
    uint32_t *s_bits = a_lds + (size_t)A.k * A.N;                         // [k][ceil(N/32)]
    const int N = A.N, k = A.k, tid = threadIdx.x;
    const int NW = (N + 31) / 32;                                         // key-bit words per polynomial (N < 32: one, partly filled)
    const size_t r = blockIdx.x;                                          // sample = (i * l + p) * (k+1) + j
    const int j = (int)(r % (size_t)(k + 1));
    const int p = (int)((r / (size_t)(k + 1)) % (size_t)A.l);
    const size_t i = r / ((size_t)(k + 1) * A.l);
    int32_t *out = A.bk + r * (size_t)(k + 1) * N;
    for (int w = tid; w < k * NW; w += 256) {
        const int c = w / NW, w0 = w % NW;
        uint32_t bits = 0;
        for (int b = 0; b < 32 && w0 * 32 + b < N; b++) bits |= (uint32_t)(A.tlwe_key[c * N + w0 * 32 + b] & 1) << b;
        s_bits[w] = bits;
    }
    for (int e = tid; e < k * N; e += 256) {
        const uint32_t v = uniform_word(1u, r * (uint64_t)k * N + e, A.seed);
        a_lds[e] = v;
        out[e] = (int32_t)v;
    }
    __syncthreads();
    const uint32_t msg = (uint32_t)(A.lwe_key[i] & 1) << (32 - (p + 1) * A.beta);
    for (int co = tid; co < N; co += 256) {
        uint32_t acc = dtot32(gaussian(2u, r * (uint64_t)N + co, A.seed) * A.bs_alpha);
        for (int c = 0; c < k; c++) {
            const uint32_t *a = a_lds + c * N;
            for (int w0 = 0; w0 < NW; w0++) {
                // Every bit position is visited and its term masked in or out: the trip count, the LDS addresses and the
                // instruction stream do not depend on the secret key bits (a `while (bits)` loop over the set bits would
                // make the kernel's duration a function of the key's Hamming weight per word).
                const uint32_t bits = s_bits[c * NW + w0];
#pragma unroll 8
                for (int b = 0; b < 32; b++) {
                    const int idx = co - (w0 * 32 + b);                   // X^m * a: coefficient co takes a[co - m], negated on wrap (m < N: the bits past N are zero)
                    const uint32_t v = a[idx & (N - 1)];
                    const uint32_t take = 0u - ((bits >> b) & 1u);        // all ones where key bit m is set
                    acc += (idx < 0 ? 0u - v : v) & take;
                }
            }
        }
        if (j == k && co == 0) acc += msg;                                // message on the body's constant term ...
        out[(size_t)k * N + co] = (int32_t)acc;
    }
    if (j < k && tid == 0) out[(size_t)j * N] = (int32_t)((uint32_t)out[(size_t)j * N] + msg);   // ... or on mask component j's
}

// keyswitch.jl:24-29: all noises first, their mean is subtracted from each
__global__ __launch_bounds__(256) void ks_noise_kernel(Args A, size_t Q)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q < Q) A.ks_noise[q] = gaussian(4u, q, A.seed) * A.ks_alpha;
}

__global__ __launch_bounds__(256) void ks_mean_kernel(Args A, size_t Q)     // one workgroup, fixed summation order
{
    __shared__ double part[256];
    double s = 0.0;
    for (size_t q = threadIdx.x; q < Q; q += 256) s += A.ks_noise[q];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) part[threadIdx.x] += part[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) A.ks_mean[0] = part[0] / (double)Q;
}

// One wave per LWE sample (i, j, h): n uniform mask words, b = message + noise + <a, s_out>   (keyswitch.jl:31-40, lwe.jl:49-55)
__global__ __launch_bounds__(256) void ks_kernel(Args A, size_t Q)
{
    const int lane = threadIdx.x & 63;
    const size_t q = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= Q) return;
    const int base1 = (1 << A.ks_log2_base) - 1;
    const int h = (int)(q % (size_t)base1) + 1;
    const int jd = (int)((q / (size_t)base1) % (size_t)A.t) + 1;
    const size_t i = q / ((size_t)base1 * A.t);
    int32_t *out = A.ks + q * (size_t)(A.n + 1);
    uint32_t dot = 0;
    for (int m = lane; m < A.n; m += 64) {
        const uint32_t v = uniform_word(3u, q * (uint64_t)A.n + m, A.seed);
        out[m] = (int32_t)v;
        if (A.lwe_key[m] & 1) dot += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
    if (lane == 0) {
        const uint32_t msg = ((uint32_t)(A.tlwe_key[i] & 1) * (uint32_t)h) << (32 - jd * A.ks_log2_base);
        out[A.n] = (int32_t)(msg + dtot32(A.ks_noise[q] - A.ks_mean[0]) + dot);
    }
}

}  // namespace keygen
