// How many 256-thread workgroups with X bytes of dynamic LDS (and V registers) does a CU really hold at once?
// Each workgroup spins for a fixed number of clock ticks; 256 * k workgroups then take ceil(k / resident) spins.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int VREGS>
__global__ __launch_bounds__(256) void spin(unsigned long long ticks, float *sink)
{
    extern __shared__ char smem[];
    float r[VREGS];
    for (int i = 0; i < VREGS; i++) r[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        for (int i = 0; i < VREGS; i++) r[i] = r[i] * 1.0001f + 0.5f;
    }
    float s = 0;
    for (int i = 0; i < VREGS; i++) s += r[i];
    if (s == 12345.678f) { smem[threadIdx.x] = 1; sink[0] = s + smem[0]; }
}
template <int VREGS>
static void run(size_t lds, int k)
{
    float *sink; hipMalloc(&sink, 4);
    hipFuncSetAttribute((const void *)spin<VREGS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const unsigned long long ticks = 100000000ull / 1000 * 2;   // 100 MHz clock: 2 ms
    hipLaunchKernelGGL(spin<VREGS>, dim3(256), dim3(256), lds, 0, ticks, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(spin<VREGS>, dim3(256 * k), dim3(256), lds, 0, ticks, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    int nb = -1; hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *)spin<VREGS>, 256, lds);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void *)spin<VREGS>);
    printf("regs %3d  LDS %6zu B  %d workgroups per CU launched: %.2f ms (= %.1f spins of 2 ms)  occupancy API says %d\n", fa.numRegs, lds, k, ms, ms / 2.0, nb);
    hipFree(sink);
}
int main()
{
    for (size_t lds : {40960ul, 49152ul, 53248ul, 54272ul, 55296ul, 65536ul}) { run<32>(lds, 3); }
    run<32>(54272, 2);
    run<150>(54272, 3); run<150>(40960, 3); run<150>(16384, 3);
    run<110>(16384, 4); run<150>(16384, 4);
    return 0;
}
