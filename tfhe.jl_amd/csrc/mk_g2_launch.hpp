// Launchers of mk_blind_rotate_kernel_g2<P, l, DIAG, RW> for the shipped 4- and 8-party sets.  Every instantiation is compiled
// in its own translation unit (mk_g2_inst.hip with -DG2_P= -DG2_L= -DG2_DG= -DG2_RW=: together two thirds of the library's
// build time, now in parallel with one another and with the engine).
#pragma once
#include <hip/hip_runtime.h>
#include "kernels_multikey.hpp"

#define TFHE_G2_LAUNCHER(P, L, DG, RW, AL) tfhe_launch_mk_g2_##P##_##L##_##DG##_##RW##_##AL
#define TFHE_G2_DECLARE(P, L, DG, RW, AL) hipError_t TFHE_G2_LAUNCHER(P, L, DG, RW, AL)(unsigned nblk, size_t lds_bytes, hipStream_t s, const MkGenArgs &ga)
// (DG = 1: the DIAG instantiation, two rotations per workgroup only; RW: rotations per workgroup; AL = 1: accumulators in LDS,
//  4 parties only)
TFHE_G2_DECLARE(4, 5, 1, 2, 1); TFHE_G2_DECLARE(4, 5, 0, 2, 1); TFHE_G2_DECLARE(4, 5, 0, 4, 1);
TFHE_G2_DECLARE(8, 8, 1, 2, 0); TFHE_G2_DECLARE(8, 8, 0, 2, 0); TFHE_G2_DECLARE(8, 8, 0, 4, 0);

// (4 parties: accumulators in LDS; 8 parties: in global memory)
inline hipError_t tfhe_launch_mk_g2(int parties, bool diag, int rw, bool acc_lds, unsigned nblk, size_t lds_bytes, hipStream_t s, const MkGenArgs &ga)
{
    if (parties == 4) {
        (void)acc_lds;
        if (diag) return TFHE_G2_LAUNCHER(4, 5, 1, 2, 1)(nblk, lds_bytes, s, ga);
        return rw == 4 ? TFHE_G2_LAUNCHER(4, 5, 0, 4, 1)(nblk, lds_bytes, s, ga) : TFHE_G2_LAUNCHER(4, 5, 0, 2, 1)(nblk, lds_bytes, s, ga);
    }
    return diag ? TFHE_G2_LAUNCHER(8, 8, 1, 2, 0)(nblk, lds_bytes, s, ga) : rw == 4 ? TFHE_G2_LAUNCHER(8, 8, 0, 4, 0)(nblk, lds_bytes, s, ga) : TFHE_G2_LAUNCHER(8, 8, 0, 2, 0)(nblk, lds_bytes, s, ga);
}
