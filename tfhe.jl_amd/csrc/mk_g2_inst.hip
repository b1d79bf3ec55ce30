// One instantiation of mk_blind_rotate_kernel_g2 and its launcher: -DG2_P=<parties> -DG2_L=<l> -DG2_DG=<0|1> -DG2_RW=<2|4> -DG2_AL=<0|1> (Makefile).
#include "mk_g2_launch.hpp"

#define G2_PASTE_(P, L, DG, RW, AL) TFHE_G2_LAUNCHER(P, L, DG, RW, AL)
#define G2_PASTE(P, L, DG, RW, AL) G2_PASTE_(P, L, DG, RW, AL)

hipError_t G2_PASTE(G2_P, G2_L, G2_DG, G2_RW, G2_AL)(unsigned nblk, size_t lds_bytes, hipStream_t s, const MkGenArgs &ga)
{
    constexpr bool DG = G2_DG != 0, AL = G2_AL != 0;
    if (lds_bytes > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute((const void *)mk_blind_rotate_kernel_g2<G2_P, G2_L, DG, G2_RW, AL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((mk_blind_rotate_kernel_g2<G2_P, G2_L, DG, G2_RW, AL>), dim3(nblk), dim3(128 * G2_RW), lds_bytes, s, ga);
    return hipGetLastError();
}
