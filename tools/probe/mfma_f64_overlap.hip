// Does the FP64 matrix pipe run beside the FP64 vector pipe on gfx950?  (the blind-rotate kernels are bound by FP64
// VALU issue; a radix pass done as v_mfma_f64_16x16x4_f64 over lane groups would also replace an LDS transposition — worth
// building only if the matrix instruction does not take the vector pipe's issue slots or multipliers.)
// Measures, with s_memtime around unrolled blocks, for 1 and 2 waves per SIMD:
//   (a) cycles per v_mfma_f64_16x16x4_f64 alone (four independent accumulators),
//   (b) cycles per group of K independent v_fma_f64 alone,
//   (c) cycles per group of one MFMA + K v_fma_f64: max(a, b) if the pipes overlap, a + b if they do not.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f64_overlap tools/probe/mfma_f64_overlap.hip && /tmp/mfma_f64_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));
#define ITERS 256

template <int K, bool MFMA>
__global__ void probe(unsigned long long *out, double *sink, double a)
{
    d4 acc[4];
    double r[16];
    for (int i = 0; i < 4; i++) acc[i] = (d4){(double)threadIdx.x, 1.0, 2.0, (double)i};
    for (int i = 0; i < 16; i++) r[i] = (double)(threadIdx.x + i);
    const double av = a + threadIdx.x, bv = a - threadIdx.x;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            if (MFMA) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(av), "v"(bv));
#pragma unroll
            for (int i = 0; i < K; i++) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[(g * K + i) & 15]) : "v"(a));
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    for (int i = 0; i < 16; i++) s += r[i];
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    if (s == 12345.0) sink[0] = s;
}

template <typename Kn>
static void run(const char *name, Kn kernel, int k, bool mfma)
{
    unsigned long long *out, h[8];
    double *sink;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 16);
    for (int waves = 1; waves <= 2; waves++) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kernel, dim3(1), dim3(64 * 4 * waves), 0, 0, out, sink, 1.0000001);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        unsigned long long hi = h[0];
        for (int w = 1; w < 4 * waves; w++) hi = h[w] > hi ? h[w] : hi;
        printf("%-28s %d wave(s) per SIMD: %7.1f cycles per group (%s%d v_fma_f64) on the SIMD\n", name, waves,
               (double)hi / (4.0 * ITERS) / waves, mfma ? "1 v_mfma_f64_16x16x4 + " : "", k);
    }
    (void)hipFree(out); (void)hipFree(sink);
}

int main()
{
    run("mfma alone", probe<0, true>, 0, true);
    run("4 fma alone", probe<4, false>, 4, false);
    run("mfma + 4 fma", probe<4, true>, 4, true);
    run("8 fma alone", probe<8, false>, 8, false);
    run("mfma + 8 fma", probe<8, true>, 8, true);
    run("16 fma alone", probe<16, false>, 16, false);
    run("mfma + 16 fma", probe<16, true>, 16, true);
    return 0;
}
