// engine_multikey.hip — multi-key: tfhe_mk_gate_nand_batch (mk_gates.jl:7-12) and the launch of its kernels
#include "engine.hpp"
#ifndef TFHE_NO_G2
#include "mk_g2_launch.hpp"
#endif

int32_t tfhe_mk_gate_nand_batch(tfhe_ctx *c, const int32_t *in0, const int32_t *in1, int32_t *out, int64_t B) try
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!in0 || !in1 || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "mk_gate_nand_batch: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->multi()) {
        const tfhe_ctx *k0 = c->kids[0];
        if (!k0->have_mk_bk || !k0->have_mk_ks) return c->set_err(TFHE_ERR_NO_KEY, "mk_gate_nand_batch: multi-key keys not loaded");
        const size_t w = (size_t)k0->mk_parties * c->P.n + 1;
        return multi_rows(c, B, [&](tfhe_ctx *k, int64_t s0, int64_t cnt) { return tfhe_mk_gate_nand_batch(k, in0 + (size_t)s0 * w, in1 + (size_t)s0 * w, out + (size_t)s0 * w, cnt); });
    }
    if (!c->have_mk_bk || !c->have_mk_ks) return c->set_err(TFHE_ERR_NO_KEY, "mk_gate_nand_batch: multi-key keys not loaded");
    // (the keyswitch loops over the bootstrapping key's parties: a keyswitch key loaded for fewer would be read past its end)
    if (c->mk_ks_parties != c->mk_parties)
        return c->set_err(TFHE_ERR_STATE, "mk_gate_nand_batch: the bootstrapping key was loaded for %d parties, the keyswitch key for %d: load both for the same parties",
                          c->mk_parties, c->mk_ks_parties);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    { const int32_t rc0 = enter_stream(c, s); if (rc0) return rc0; }
    const int NP = c->mk_parties, n = c->P.n, nw = NP * n + 1, Nn = c->P.N /* (1024 in every tuned branch below) */, ew = NP * Nn + 1;
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
    c->map_cur->pending = true;
    const int32_t *d_gate = (const int32_t *)c->map.p;
    const uint8_t *d_kind = (const uint8_t *)(d_gate + B);
    next_timing_slot(c);
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    // mk_gate_nand prologue (mk_gates.jl:8-10) = the NAND affine form over P*n+1 words, then mod-switch
    rc = launch_prologue(c, (size_t)B, (const int32_t *)c->io[0].p, (const int32_t *)c->io[1].p, nullptr, d_gate, d_gate, d_kind, NP * n, s);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    MkBrArgs a;
    rc = prepare_diag(c, (size_t)B, s, a.diag);
    if (rc) return rc;
    const bool dg = c->measure_margin;
    a.bara = (const int32_t *)c->bara.p; a.bk = c->d_mk_bk; a.ext = (int32_t *)c->ext.p; a.T = c->T; a.g = c->g;
    a.n = n; a.mu = (int32_t)(1u << 29); a.R = (int32_t)B;
    a.prio_steps = (int32_t)((int64_t)NP * n * c->br_prio_pct / 100);
    const size_t lds = (size_t)(NP + 1) * kImg * 4 + (kXchElems + 64) * sizeof(cplx);
    // 2 parties with l = 4 (mktfhe_parameters_2party, mk_api.jl:4-10): the tuned two-wave kernel; any other shape, and option
    // mk_general, the any-party kernel (round 3's one-wave 2-party kernel is gone: the any-party kernel is the cross-check)
    const bool special = (NP == 2 && c->P.bs_l == 4 && !c->mk_force_general);
    if (c->anyn()) {
        // any N, any number of parties, any l (kernels_anyn.hpp): one workgroup per rotation, accumulators in global memory
        const int M = Nn / 2;
        anyn::Args g;
        g.diag = a.diag; g.bara = a.bara; g.bk = a.bk; g.ext = a.ext; g.g = c->g; g.n = n; g.mu = a.mu; g.K1 = NP + 1; g.L = c->P.bs_l; g.R = (int32_t)B;
        g.log2N = ilog2i(Nn); g.parties = NP;
        g.wtab = c->d_anyn_tab; g.twist = c->d_anyn_tab + M;
        HIP_TRY(c, c->mk_acc.reserve((size_t)B * (NP + 1) * Nn * sizeof(int32_t)));
        g.acc = (int32_t *)c->mk_acc.p;
        const bool fits = anyn::lds_bytes(Nn, 3) <= 160 * 1024;
        const bool spec_lds = c->anyn_spec < 0 ? fits : (c->anyn_spec == 0 && fits);
        g.spec_g = nullptr;
        if (!spec_lds) {
            HIP_TRY(c, c->spec.reserve((size_t)B * 3 * (M > 0 ? M : 1) * sizeof(cplx)));
            g.spec_g = (cplx *)c->spec.p;
        }
        const size_t ldsa = anyn::lds_bytes(Nn, spec_lds ? 3 : 0);
        const unsigned nt = (unsigned)anyn::threads_for(Nn);
        if (dg) {
            if (ldsa > 64 * 1024) LDS_TRY(c, ldsa, anyn::mk_blind_rotate_kernel<true>);
            hipLaunchKernelGGL((anyn::mk_blind_rotate_kernel<true>), dim3((unsigned)B), dim3(nt), ldsa, s, g);
        } else {
            if (ldsa > 64 * 1024) LDS_TRY(c, ldsa, anyn::mk_blind_rotate_kernel<false>);
            hipLaunchKernelGGL((anyn::mk_blind_rotate_kernel<false>), dim3((unsigned)B), dim3(nt), ldsa, s, g);
        }
        name_kernel(c, spec_lds ? "mk_blind_rotate_kernel_anyn(N=%d,P=%d,l=%d)" : "mk_blind_rotate_kernel_anyn(N=%d,P=%d,l=%d,spec=global)", Nn, NP, c->P.bs_l);
    } else if (special) {
        // two waves per rotation: acc[3][N] | xch[2] | second hand-off slot [M] | tw2   (39.4 KB: four workgroups per CU)
        // mk_rw rotations per workgroup in lockstep (2: default: 78.8 KB, two workgroups per CU; 1: 39.4 KB, four; DIAG: 1)
        const int rw = dg ? 1 : c->mk_rw ? c->mk_rw : ((size_t)B <= (size_t)c->cu_count ? 1 : 2);
        const size_t lds2 = (size_t)rw * ((NP + 1) * kImg * 4 + (2 * kXchElems + kM) * sizeof(cplx)) + 64 * sizeof(cplx) + 64;      // (+ the hand-off words of the pairs)
        const unsigned nblk = (unsigned)((B + rw - 1) / rw);
        a.R = (int32_t)B;
#define LAUNCH_MK2(LL, DG, RWV)                                                                                    \
        do {                                                                                                       \
            if (lds2 > 64 * 1024)                                                                                  \
                LDS_TRY(c, lds2, mk_blind_rotate_kernel_w2<LL, DG, RWV>); \
            hipLaunchKernelGGL((mk_blind_rotate_kernel_w2<LL, DG, RWV>), dim3(nblk), dim3(128 * RWV), lds2, s, a);  \
        } while (0)
        if (dg) LAUNCH_MK2(4, true, 1);
        else if (rw == 2) LAUNCH_MK2(4, false, 2);
        else LAUNCH_MK2(4, false, 1);
#undef LAUNCH_MK2
        name_kernel(c, "mk_blind_rotate_kernel_w2<%d>", c->P.bs_l);
#ifndef TFHE_NO_G2      // (-DTFHE_NO_G2: quick development builds without the many-party two-wave kernel, 1 instead of 5 minutes)
    } else if (!c->mk_force_general && c->mkg_variant != 1 && ((NP == 4 && c->P.bs_l == 5) || (NP == 8 && c->P.bs_l == 8))) {
        // the shipped 4- and 8-party sets (mk_api.jl:16-34): compile-time (parties, l), two waves per rotation at two waves per
        // SIMD, accumulators in global memory.  LDS: two transposition buffers per rotation and the pass-B twiddle table;
        // two rotations per workgroup in lockstep (a single rotation gets a padding partner)
        MkGenArgs ga;
        ga.diag = a.diag; ga.R = (int32_t)B; ga.bara = a.bara; ga.bk = a.bk; ga.ext = a.ext; ga.T = a.T; ga.g = a.g; ga.n = n; ga.mu = a.mu; ga.parties = NP; ga.L = c->P.bs_l;
        ga.prio_steps = a.prio_steps;
        // rotations per workgroup, in lockstep (they share their key fetches): "mkg_rw" 2 | 4, default 4 = one workgroup of
        // eight waves per CU.  The 8-party key is 4.7 GB as spectra: with pairs the launch moves 2.2 TB beyond L2 (4.9 TB/s,
        // L2 hit 59 %, profiles/r03/r03l_mk8: every pair streams the whole key for itself) and takes 450 ms; four rotations
        // per workgroup halve that traffic: 388 ms.  4 parties: 73.1 vs 73.9 ms.  The DIAG instantiation exists for pairs only.
        // (Measured dead end: pacing the workgroups of an XCD — a counter per XCD, one lane per workgroup waiting, bounded,
        //  until its XCD's workgroups have all finished the step, so that they share key lines in their L2 — costs more in
        //  waiting for the slowest of 32 than it saves: 8 parties 424 vs 403 ms, 4 parties 94 vs 80 ms on one device.)
        // (up to two rotations per CU the pairs win: 4 parties 54 vs 69 ms at 512 rotations, 62 vs 79 ms for a single gate;
        //  8 parties 298 vs 374 ms at 256 — profiles/r03/r03j_*)
        // 4 parties: the five accumulator images (21.8 KB per rotation) fit LDS beside the transposition buffers at four
        // rotations per CU — 81 408 B per pair of rotations = 40 LDS granules of 2 KB, two pairs or one group of four per CU —
        // so the step needs neither the trip to L2 nor the workgroup-scope fence (round 4: 66.2 vs 73.6 ms; at 8 parties nine
        // images do not fit and the accumulators stay in global memory)
        const bool acc_lds = NP == 4;
        const int rw = dg ? 2 : (c->mkg_rw == 2 || c->mkg_rw == 4) ? c->mkg_rw : ((size_t)B <= 2 * (size_t)c->cu_count ? 2 : 4);
        const size_t ldsg2 = (size_t)rw * 2 * kXchElems * sizeof(cplx) + 64 * sizeof(cplx) + (acc_lds ? (size_t)rw * (NP + 1) * kImg * sizeof(int32_t) : 0);
        const unsigned nblk = (unsigned)((B + rw - 1) / rw);
        ga.acc = nullptr;
        if (!acc_lds) {
            HIP_TRY(c, c->mk_acc.reserve((size_t)nblk * rw * (NP + 1) * kImg * sizeof(int32_t)));
            ga.acc = (int32_t *)c->mk_acc.p;
        }
        HIP_TRY(c, tfhe_launch_mk_g2(NP, dg, rw, acc_lds, nblk, ldsg2, s, ga));
        name_kernel(c, acc_lds ? "mk_blind_rotate_kernel_g2<%d,%d,acc=lds>" : "mk_blind_rotate_kernel_g2<%d,%d>", NP, c->P.bs_l);
#endif
    } else {
        MkGenArgs ga;
        ga.diag = a.diag; ga.R = (int32_t)B; ga.bara = a.bara; ga.bk = a.bk; ga.ext = a.ext; ga.T = a.T; ga.g = a.g; ga.n = n; ga.mu = a.mu; ga.parties = NP; ga.L = c->P.bs_l;
        // the kernel needs a whole SIMD's registers, so a CU holds four waves whatever the grouping: as many rotations per
        // workgroup (in lockstep, sharing their key fetches) as fit in LDS, four at most
        // accumulators in global memory: LDS holds only the transposition buffer, eight waves fit a CU whatever P is
        const bool accg = c->mkg_acc < 0 ? NP > 4 : c->mkg_acc != 0;
        const size_t lds_rot = accg ? (kXchElems + 64) * sizeof(cplx) : lds;
        // (two rotations per workgroup: 82 vs 86 ms with four or one at 4 parties, 476 vs 481 / 765 at 8 — round 2; the three- and
        //  four-rotation instantiations are gone)
        int rw = (int)std::min<size_t>(2, (160 * 1024) / lds_rot);
        if (c->mkg_rw == 1 || B < 2 || rw < 1 || dg) rw = 1;       // (the DIAG instantiations exist for single rotations only)
        const size_t ldsg = (size_t)rw * lds_rot;
        const unsigned nblk = (unsigned)((B + rw - 1) / rw);
        ga.acc = nullptr;
        ga.prio_steps = 0;
        if (accg) {
            HIP_TRY(c, c->mk_acc.reserve((size_t)nblk * rw * (NP + 1) * kImg * sizeof(int32_t)));
            ga.acc = (int32_t *)c->mk_acc.p;
        }
#define LAUNCH_MKG(DG, RWV)                                                                                        \
        do {                                                                                                       \
            if (ldsg > 64 * 1024)                                                                                  \
                LDS_TRY(c, ldsg, mk_blind_rotate_kernel_general<DG, RWV, false>); \
            if (accg) hipLaunchKernelGGL((mk_blind_rotate_kernel_general<DG, RWV, true>), dim3(nblk), dim3(64 * RWV), ldsg, s, ga); \
            else hipLaunchKernelGGL((mk_blind_rotate_kernel_general<DG, RWV, false>), dim3(nblk), dim3(64 * RWV), ldsg, s, ga); \
        } while (0)
        if (dg) LAUNCH_MKG(true, 1);
        else if (rw == 2) LAUNCH_MKG(false, 2);
        else LAUNCH_MKG(false, 1);
#undef LAUNCH_MKG
        name_kernel(c, accg ? "mk_blind_rotate_kernel_general(P=%d,L=%d,acc=global)" : "mk_blind_rotate_kernel_general(P=%d,L=%d)", NP, c->P.bs_l);
    }
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    rc = launch_mk_keyswitch(c, (size_t)B, d_gate, (int32_t *)c->io[3].p, s);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, bytes, hipMemcpyDeviceToHost, s));
    rc = leave_stream(c, s);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(s));
    commit_timing_slot(c);
    c->last_rotations = B;
    return TFHE_OK;
}
ABI_CATCH(c, "tfhe_mk_gate_nand_batch")
