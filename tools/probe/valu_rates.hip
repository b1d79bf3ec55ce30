// Issue cost of the vector instructions the blind-rotate kernels are made of, on gfx950: cycles per wave-instruction for
// ONE wave on a SIMD and for two, three and four waves sharing it (s_memtime around an unrolled block of independent instructions).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/probe/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 16        // independent chains
#define ITERS 64

#define BODY(ASM, CONSTR_OUT, CONSTR_IN, T, INIT)                                                          \
    T r[REP];                                                                                              \
    for (int i = 0; i < REP; i++) r[i] = INIT;                                                             \
    __builtin_amdgcn_s_barrier();                                                                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                            \
    for (int it = 0; it < ITERS; it++) {                                                                   \
        _Pragma("unroll") for (int i = 0; i < REP; i++) ASM;                                               \
    }                                                                                                      \
    asm volatile("s_nop 0" ::: "memory");                                                                  \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                            \
    T s = 0;                                                                                               \
    for (int i = 0; i < REP; i++) s += r[i];                                                               \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                              \
    if (s == (T)12345) sink[0] = (double)s;

__global__ void k_add_f64(unsigned long long *out, double *sink, double a)
{
    BODY(asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[i]) : "v"(a)), , , double, (double)(threadIdx.x + i))
}
__global__ void k_mul_f64(unsigned long long *out, double *sink, double a)
{
    BODY(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[i]) : "v"(a)), , , double, (double)(threadIdx.x + i))
}
__global__ void k_fma_f64(unsigned long long *out, double *sink, double a)
{
    BODY(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a)), , , double, (double)(threadIdx.x + i))
}
__global__ void k_cvt_f64_i32(unsigned long long *out, double *sink, double a)
{
    int src = (int)threadIdx.x;
    BODY(asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(r[i]) : "v"(src)), , , double, (double)(threadIdx.x + i))
}
__global__ void k_cvt_i32_f64(unsigned long long *out, double *sink, double a)
{
    int q[REP];
    BODY(asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(q[i]) : "v"(a)), , , double, (double)(threadIdx.x + i))
    if (q[0] == 77 && q[REP - 1] == 78) sink[1] = 1.0;
}
__global__ void k_bfe_i32(unsigned long long *out, double *sink, double a)
{
    BODY(asm volatile("v_bfe_i32 %0, %0, 3, 7" : "+v"(r[i])), , , int, (int)(threadIdx.x * 977 + i))
}
__global__ void k_and_b32(unsigned long long *out, double *sink, double a)
{
    BODY(asm volatile("v_and_b32 %0, 0x7f7f7f7f, %0" : "+v"(r[i])), , , int, (int)(threadIdx.x * 977 + i))
}
__global__ void k_add_u32(unsigned long long *out, double *sink, double a)
{
    BODY(asm volatile("v_add_u32 %0, 3, %0" : "+v"(r[i])), , , int, (int)(threadIdx.x * 977 + i))
}
__global__ void k_mov_b32(unsigned long long *out, double *sink, double a)
{
    int src = (int)threadIdx.x;
    BODY(asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(src)), , , int, (int)(threadIdx.x * 977 + i))
}
__global__ void k_fma_f32(unsigned long long *out, double *sink, double a)
{
    float af = (float)a;
    BODY(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(af)), , , float, (float)(threadIdx.x + i))
}
__global__ void k_pk_fma_f32(unsigned long long *out, double *sink, double a)
{
    BODY(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(a)), , , double, (double)(threadIdx.x + i))
}

template <typename K>
static void run(const char *name, K kernel)
{
    unsigned long long *out, h[16];
    double *sink;
    (void)hipMalloc(&out, 128); (void)hipMalloc(&sink, 16);
    for (int waves = 1; waves <= 4; waves++) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kernel, dim3(1), dim3(64 * 4 * waves), 0, 0, out, sink, 1.0000001);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, out, 128, hipMemcpyDeviceToHost);
        unsigned long long lo = h[0], hi = h[0];
        for (int w = 1; w < 4 * waves; w++) { lo = h[w] < lo ? h[w] : lo; hi = h[w] > hi ? h[w] : hi; }
        // the SIMD's arbiter favours its oldest wave: the fastest wave shows what one wave can issue, the slowest what the pipe sustains
        printf("%-14s %d wave(s) per SIMD: fastest wave %6.2f, slowest %6.2f cycles per wave-instruction -> %5.2f cycles per instruction on the SIMD\n",
               name, waves, (double)lo / (REP * ITERS), (double)hi / (REP * ITERS), (double)hi / (REP * ITERS) / waves);
    }
    (void)hipFree(out); (void)hipFree(sink);
}

int main()
{
    // s_memtime counts at the shader clock on gfx950? (the engine's DIAG instantiations assume so: s_memtime / s_memrealtime x 100 MHz)
    run("v_add_f64", k_add_f64);
    run("v_mul_f64", k_mul_f64);
    run("v_fma_f64", k_fma_f64);
    run("v_cvt_f64_i32", k_cvt_f64_i32);
    run("v_cvt_i32_f64", k_cvt_i32_f64);
    run("v_bfe_i32", k_bfe_i32);
    run("v_and_b32", k_and_b32);
    run("v_add_u32", k_add_u32);
    run("v_mov_b32", k_mov_b32);
    run("v_fma_f32", k_fma_f32);
    run("v_pk_fma_f32", k_pk_fma_f32);
    return 0;
}
