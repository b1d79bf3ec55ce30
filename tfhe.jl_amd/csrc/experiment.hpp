// experiment.hpp — development builds declare themselves.
//
// The shipped library is built from these sources with no -D switch that changes a kernel or a dispatch decision.  What a
// development build may switch on — in-kernel phase stamps (TFHE_STAMP: `make stamp`, tools/phase_profile.py), a build without
// the many-party kernels (TFHE_NO_G2), host-time printing and environment-variable overrides (TFHE_DEBUG_HOSTTIME,
// TFHE_DEBUG_LIFETIMES, TFHE_ANYN_THREADS) — compiles only under -DTFHE_EXPERIMENT, and such a library answers
// tfhe_abi_version() with the NEGATIVE of its ABI version: tfhe.jl_amd/_lib.py and julia/TFHEMI355X refuse to load it unless
// TFHE_MI355X_ALLOW_EXPERIMENT=1 says the caller knows what it is measuring.  (Rounds 3-5 kept timing-only ablations of the
// kernels — TFHE_ABL_*, TFHE_NO_TANFORM, TFHE_V3_KMID, ... — as #ifdefs in the kernel sources; their numbers are in
// profiles/DESIGN_r01-r04_history.md and profiles/r05/, the switches are gone.)
#pragma once
#if (defined(TFHE_STAMP) || defined(TFHE_NO_G2)) && !defined(TFHE_EXPERIMENT)
#error "TFHE_STAMP / TFHE_NO_G2 are development switches: build with -DTFHE_EXPERIMENT (the library then reports a negative ABI version)"
#endif
#ifdef TFHE_EXPERIMENT
#define TFHE_ABI_VERSION_REPORTED (-(TFHE_MI355X_ABI_VERSION))
#else
#define TFHE_ABI_VERSION_REPORTED (TFHE_MI355X_ABI_VERSION)
#endif
