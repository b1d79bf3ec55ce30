// engine_dispatch.hip — dispatch: which blind-rotate / keyswitch kernel a batch takes (by parameter set and batch size) and its launch
#define TFHE_EMIT_KEYSWITCH_KERNELS
#include "engine.hpp"

#include <map>
#include <mutex>
#include <functional>

int32_t ensure_dyn_lds(tfhe_ctx *c, const void *fn, size_t bytes, const char *what)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> have;
    std::lock_guard<std::mutex> lk(mu);
    size_t &cur = have[std::make_pair(c->device, fn)];
    if (cur >= bytes) return TFHE_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return c->set_err(TFHE_ERR_DEVICE, "hipFuncSetAttribute(%s, %zu bytes of LDS) failed: %s", what, bytes, hipGetErrorString(e));
    cur = bytes;
    return TFHE_OK;
}

// ---- launch helpers ------------------------------------------------------------------------------
int32_t prepare_diag(tfhe_ctx *c, size_t R, hipStream_t s, DiagArgs &d)
{
    d.margin_bits = nullptr; d.clk = nullptr; d.phase = nullptr;
    c->diag_rows = 0;
    if (!c->measure_margin) return TFHE_OK;
    HIP_TRY(c, c->diag.reserve(R * 24 + 64 * 8));
    HIP_TRY(c, hipMemsetAsync(c->diag.p, 0, R * 24 + 64 * 8, s));
    d.margin_bits = (unsigned long long *)c->diag.p;
    d.clk = d.margin_bits + R;
    d.phase = d.clk + 2 * R;
    c->diag_rows = R;
    return TFHE_OK;
}

// Tuned kernels are instantiated for the decomposition lengths the shipped parameter sets use: l = 2 (tfhe_parameters_80,
// api.jl:30-52) and l = 3 (tfhe_parameters_128, api.jl:55-69; BASELINE config 4b), with either mask size at N = 1024.
// Every other set the reference would accept runs on blind_rotate_kernel_general.
#define BR_CASES(LAUNCH)                                                                                           \
    switch (c->P.bs_l) {                                                                                           \
    case 2: LAUNCH(2); break;                                                                                      \
    case 3: LAUNCH(3); break;                                                                                      \
    default: return c->set_err(TFHE_ERR_STATE, "blind rotate: no tuned kernel for bs_l = %d", c->P.bs_l);          \
    }
// ... the one- and two-waves-per-rotation kernels also exist with the decomposition length as a run-time value (L = 0)
#define BR_CASES_ANY_L(LAUNCH)                                                                                     \
    switch (c->br_rt_l ? 0 : c->P.bs_l) {                                                                          \
    case 2: LAUNCH(2); break;                                                                                      \
    case 3: LAUNCH(3); break;                                                                                      \
    default: LAUNCH(0); break;                                                                                     \
    }

void name_kernel(tfhe_ctx *c, const char *fmt, ...)
{
    char buf[128];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    c->last_kernel = buf;
}

// Blind rotation of rotations [first, first + R) of the batch (rows of the bara / ext workspaces): picks the kernel for a
// batch of R rotations, launches it on `s` and names it.
static int32_t launch_blind_rotate_part(tfhe_ctx *c, size_t first, size_t R, int32_t mu, hipStream_t s, const DiagArgs &diag, int k2_kind = -1 /* k = 2: 0 = one wave per rotation, 1 = three, -1 = by batch size */)
{
    BrArgs a;
    a.diag = diag;
    if (diag.margin_bits) { a.diag.margin_bits += first; a.diag.clk += 2 * first; }
    const bool dg = c->measure_margin;
    a.bara = (const int32_t *)c->bara.p + first * (size_t)(c->P.n + 1);
    a.bk = c->d_bk;
    a.ext = (int32_t *)c->ext.p + first * ((size_t)c->P.k * c->P.N + 1);
    a.T = c->T;
    a.g = c->g;
    a.n = c->P.n;
    a.mu = mu;
    a.prio_steps = (int32_t)((int64_t)c->P.n * c->br_prio_pct / 100);
    a.R = (int32_t)R;
    const int L = c->P.bs_l;
    a.l = L;
    if (c->anyn()) {
        // any power-of-two N, any k, any l: one workgroup per rotation, in-LDS mixed-radix transforms (kernels_anyn.hpp)
        const int K1 = c->P.k + 1, N = c->P.N, M = N / 2;
        anyn::Args g;
        g.diag = a.diag; g.bara = a.bara; g.bk = a.bk; g.ext = a.ext; g.g = c->g; g.n = a.n; g.mu = mu; g.K1 = K1; g.L = L; g.R = (int32_t)R;
        g.log2N = ilog2i(N); g.parties = 1;
        g.wtab = c->d_anyn_tab; g.twist = c->d_anyn_tab + M;
        HIP_TRY(c, c->mk_acc.reserve((first + R) * (size_t)K1 * N * sizeof(int32_t)));
        g.acc = (int32_t *)c->mk_acc.p + first * (size_t)K1 * N;
        // spectrum accumulators in LDS when the whole workgroup fits a CU's 160 KB, in global memory otherwise
        const bool fits = anyn::lds_bytes(N, K1) <= 160 * 1024;
        const bool spec_lds = c->anyn_spec < 0 ? fits : (c->anyn_spec == 0 && fits);
        g.spec_g = nullptr;
        if (!spec_lds) {
            HIP_TRY(c, c->spec.reserve((first + R) * (size_t)K1 * (M > 0 ? M : 1) * sizeof(cplx)));
            g.spec_g = (cplx *)c->spec.p + first * (size_t)K1 * (M > 0 ? M : 1);
        }
        const size_t ldsa = anyn::lds_bytes(N, spec_lds ? K1 : 0);
        const unsigned nt = (unsigned)anyn::threads_for(N);
        if (dg) {
            if (ldsa > 64 * 1024) LDS_TRY(c, ldsa, anyn::blind_rotate_kernel<true>);
            hipLaunchKernelGGL((anyn::blind_rotate_kernel<true>), dim3((unsigned)R), dim3(nt), ldsa, s, g);
        } else {
            if (ldsa > 64 * 1024) LDS_TRY(c, ldsa, anyn::blind_rotate_kernel<false>);
            hipLaunchKernelGGL((anyn::blind_rotate_kernel<false>), dim3((unsigned)R), dim3(nt), ldsa, s, g);
        }
        HIP_TRY(c, hipGetLastError());
        name_kernel(c, spec_lds ? "blind_rotate_kernel_anyn(N=%d,k=%d,l=%d)" : "blind_rotate_kernel_anyn(N=%d,k=%d,l=%d,spec=global)", N, c->P.k, L);
        return TFHE_OK;
    }
    if (c->n512()) {
        // N = 512, k = 1: blind_rotate_kernel_v3's design with four points per lane (kernels_n512.hpp), three waves per SIMD;
        // four rotations per workgroup in lockstep once the batch fills the chip (option n512_rw)
        N512Args b;
        b.diag = a.diag; b.bara = a.bara; b.bk = a.bk; b.ext = a.ext; b.g = c->g; b.n = a.n; b.mu = mu; b.R = (int32_t)R; b.l = L;
        b.prio_steps = a.prio_steps;
        b.tw1 = c->d_tables + kN512TableOffset; b.tw2q = c->d_tables + kH2TableOffset + 512; b.tw3q = b.tw2q + 64;
        // two waves per rotation (wave c owns polynomial c) while the batch leaves SIMDs idle on the one-wave kernel: option
        // "n512_w2" (-1: up to 6 rotations per CU, 0: never, 1: always).  One device, tfhe_parameters_80 with N = 512
        // (profiles/r05/r05k_n512_timing.txt): 1 rotation 1.13 vs 1.98 ms, 1024: 1.94 vs 2.58, 1536: 2.71 vs 2.96, 2048: 3.92 vs 3.44
        if (c->n512_w2 == 1 || (c->n512_w2 < 0 && R <= 6 * (size_t)c->cu_count)) {
            const size_t ldsw = kN512W2LdsBytes;
#define LAUNCH_N512W2(LL)                                                                                          \
            if (dg) hipLaunchKernelGGL((blind_rotate_kernel_n512w2<LL, true>), dim3((unsigned)R), dim3(128), ldsw, s, b);        \
            else hipLaunchKernelGGL((blind_rotate_kernel_n512w2<LL, false>), dim3((unsigned)R), dim3(128), ldsw, s, b)
            BR_CASES_ANY_L(LAUNCH_N512W2)
#undef LAUNCH_N512W2
            HIP_TRY(c, hipGetLastError());
            if ((L == 2 || L == 3) && !c->br_rt_l) name_kernel(c, "blind_rotate_kernel_n512w2<%d>", L);
            else name_kernel(c, "blind_rotate_kernel_n512w2<0>(l=%d)", L);
            return TFHE_OK;
        }
        const bool group = !dg && (c->n512_rw == 4 || (c->n512_rw == 0 && R >= 8 * (size_t)c->cu_count));      // (2048 rotations: 3.44 vs 3.82 ms, 3072: 4.68 vs 5.11)
        const size_t lds5 = (size_t)(group ? 4 : 1) * kN512LdsBytes;
#define LAUNCH_N512(LL)                                                                                            \
        if (dg) hipLaunchKernelGGL((blind_rotate_kernel_n512<LL, true, 1>), dim3((unsigned)R), dim3(64), lds5, s, b);            \
        else if (group) hipLaunchKernelGGL((blind_rotate_kernel_n512<LL, false, 4>), dim3((unsigned)((R + 3) / 4)), dim3(256), lds5, s, b); \
        else hipLaunchKernelGGL((blind_rotate_kernel_n512<LL, false, 1>), dim3((unsigned)R), dim3(64), lds5, s, b)
        BR_CASES_ANY_L(LAUNCH_N512)
#undef LAUNCH_N512
        HIP_TRY(c, hipGetLastError());
        if ((L == 2 || L == 3) && !c->br_rt_l) name_kernel(c, group ? "blind_rotate_kernel_n512<%d,rw4>" : "blind_rotate_kernel_n512<%d>", L);
        else name_kernel(c, group ? "blind_rotate_kernel_n512<0,rw4>(l=%d)" : "blind_rotate_kernel_n512<0>(l=%d)", L);
        return TFHE_OK;
    }
    const bool tuned = c->P.N == kN2 ? (c->P.k == 1 && L == 3) : c->P.k == 1 ? true : (c->P.k == 2 && (L == 2 || L == 3));
    if (!tuned || c->br_general) {
        // any (k <= 4, l, N): one wave per rotation, accumulator images in global memory, spectrum accumulators in LDS
        const int K1 = c->P.k + 1, H = c->P.N / kN;
        const size_t img = (size_t)kMir + c->P.N;
        BrGenArgs g;
        g.diag = a.diag; g.bara = a.bara; g.bk = a.bk; g.ext = a.ext; g.g = c->g; g.n = a.n; g.mu = mu; g.K1 = K1; g.L = L; g.R = (int32_t)R;
        HIP_TRY(c, c->mk_acc.reserve((first + R) * K1 * img * sizeof(int32_t)));
        g.acc = (int32_t *)c->mk_acc.p + first * K1 * img;
        g.tw1f = c->P.N == kN2 ? (const cplx *)(c->d_tables + kTableElems) : c->T.tw1f;
        g.tw2 = c->T.tw2;
        const size_t ldsg = (kXchElems + 64 + (size_t)K1 * H * kM) * sizeof(cplx);
#define LAUNCH_GEN(NB, DG)                                                                                         \
        do {                                                                                                       \
            if (ldsg > 64 * 1024)                                                                                  \
                LDS_TRY(c, ldsg, blind_rotate_kernel_general<NB, DG>); \
            hipLaunchKernelGGL((blind_rotate_kernel_general<NB, DG>), dim3((unsigned)R), dim3(64), ldsg, s, g);     \
        } while (0)
        if (c->P.N == kN2) { if (dg) LAUNCH_GEN(32, true); else LAUNCH_GEN(32, false); }
        else { if (dg) LAUNCH_GEN(16, true); else LAUNCH_GEN(16, false); }
#undef LAUNCH_GEN
        HIP_TRY(c, hipGetLastError());
        name_kernel(c, "blind_rotate_kernel_general(N=%d,k=%d,l=%d)", c->P.N, c->P.k, L);
        return TFHE_OK;
    }
    if (c->P.N == kN2) {
        // BASELINE config 4b's shape (k = 1, l = 3).  n2048_rw rotations per workgroup in lockstep (2: default; 1: one rotation
        // per workgroup, up to one rotation per CU: the pair would leave half the CUs idle).  Four per workgroup — the whole CU
        // in phase — were measured slower: 47.4 vs 44.5 ms per 4096 rotations (profiles/r04/r04a_4b.jsonl).
        Br2048Args b;
        b.diag = a.diag; b.bara = a.bara; b.bk = a.bk; b.ext = a.ext; b.tw1f2 = c->d_tables + kTableElems; b.tw2 = c->T.tw2; b.g = c->g; b.n = a.n; b.mu = mu; b.prio_steps = a.prio_steps;
        b.R = (int32_t)R;
        const int rw = dg ? 1 : c->n2048_rw ? c->n2048_rw : (R <= (size_t)c->cu_count ? 1 : 2);     // (the DIAG instantiation exists for single rotations only)
        const size_t ldsb = (size_t)rw * (2 * kImg2 * 4 + 2 * kXchElems * sizeof(cplx)) + 64 * sizeof(cplx) + 64;      // (+ the hand-off words of the pairs)
        const unsigned nblk = (unsigned)((R + rw - 1) / rw);
#define LAUNCH_2048(DG, RWV)                                                                                       \
        do {                                                                                                       \
            if (ldsb > 64 * 1024)                                                                                  \
                LDS_TRY(c, ldsb, blind_rotate_kernel_n2048x<3, DG, RWV>); \
            hipLaunchKernelGGL((blind_rotate_kernel_n2048x<3, DG, RWV>), dim3(nblk), dim3(128 * RWV), ldsb, s, b); \
        } while (0)
        if (dg) LAUNCH_2048(true, 1);
        else if (rw == 2) LAUNCH_2048(false, 2);
        else LAUNCH_2048(false, 1);
#undef LAUNCH_2048
        HIP_TRY(c, hipGetLastError());
        name_kernel(c, "blind_rotate_kernel_n2048x<%d,rw%d>", L, rw);
        return TFHE_OK;
    }
    if (c->P.k == 2) {
        const size_t ldsk = kK2LdsBytes;      // (no mirror blocks in this kernel: seven rotations per CU)
        // Up to seven rotations per workgroup in lockstep = one workgroup per CU: the k = 2 key is 73.7 MB of spectra, 144 KB
        // per step and rotation, and independent waves stream it through the 4 MB L2 of their XCD at 77 % hits (58 GB beyond
        // L2 per 4096 rotations, VALU busy 0.46: profiles/r03/r03k2_*); in lockstep the waves of a CU share every key line in
        // its L1 and a full round of 1792 rotations takes 10.15 ms (0.81 of the roofline).  A partly filled round takes
        // almost as long as a full one, so the batch is dealt out in ceil(R / 1792) EQUALLY full rounds: every workgroup
        // gets floor or ceil of R / (rounds x CUs) rotations, its other waves idle at the barriers (4096 rotations: rounds
        // of 6, 5 and 5 per CU).  Option k2_rw: 0 / 7 = this rule, 1 = single-rotation workgroups.
        // (Round 4, measured dead end: groups of THREE in lockstep, two such workgroups per CU, handed out by the dispatcher as
        //  slots free up — no rounds, six rotations per CU: 1792 rotations 15.8 vs 11.4 ms, 4096: 30.8 vs 29.9, 7168: 49.9 vs 43.5,
        //  16384: 105.5 vs 100.1; profiles/r04/r04b_k2.jsonl)
        const size_t cus = (size_t)c->cu_count;
        // three waves per rotation (blind_rotate_kernel_k2w3: wave c owns polynomial c): up to two rotations per CU, where the
        // one-wave kernel would keep one SIMD in four busy — and the last round of a larger batch (k2_partition)
        const bool w3 = k2_kind >= 0 ? k2_kind == 1 : (c->k2_w3 == 1 || (c->k2_w3 < 0 && R <= 2 * cus));
        if (w3) {
            const size_t ldsw = kK2W3LdsBytes;
#define LAUNCH_K2W3(LL)                                                                                            \
            do {                                                                                                   \
                if (dg) { LDS_TRY(c, ldsw, blind_rotate_kernel_k2w3<LL, true>); hipLaunchKernelGGL((blind_rotate_kernel_k2w3<LL, true>), dim3((unsigned)R), dim3(192), ldsw, s, a); } \
                else { LDS_TRY(c, ldsw, blind_rotate_kernel_k2w3<LL, false>); hipLaunchKernelGGL((blind_rotate_kernel_k2w3<LL, false>), dim3((unsigned)R), dim3(192), ldsw, s, a); } \
            } while (0)
            BR_CASES(LAUNCH_K2W3)
#undef LAUNCH_K2W3
            HIP_TRY(c, hipGetLastError());
            name_kernel(c, "blind_rotate_kernel_k2w3<%d>", L);
            return TFHE_OK;
        }
        const bool grouped = !dg && (c->k2_rw == 7 || c->k2_rw == 0);     // (never slower than single-rotation workgroups: 6.7 vs 6.9 ms at 64 rotations, 6.8 vs 7.6 at 512)
        if (grouped) {
            const size_t rounds = (R + 7 * cus - 1) / (7 * cus);
            const size_t G = std::min(R, rounds * cus);                 // workgroups; fewer than one per CU only for tiny batches
            a.grp_q = (int32_t)(R / G);
            a.grp_big = (int32_t)(R % G);
#define LAUNCH_K2(LL)                                                                                              \
            do {                                                                                                   \
                LDS_TRY(c, (7 * ldsk), blind_rotate_kernel_k2<LL, false, 7>); \
                hipLaunchKernelGGL((blind_rotate_kernel_k2<LL, false, 7>), dim3((unsigned)G), dim3(448), 7 * ldsk, s, a); \
            } while (0)
            BR_CASES(LAUNCH_K2)
#undef LAUNCH_K2
            HIP_TRY(c, hipGetLastError());
            name_kernel(c, "blind_rotate_kernel_k2<%d,rw7>", L);
            return TFHE_OK;
        }
#define LAUNCH_K2(LL)                                                                                              \
        if (dg) hipLaunchKernelGGL((blind_rotate_kernel_k2<LL, true>), dim3((unsigned)R), dim3(64), ldsk, s, a);  \
        else hipLaunchKernelGGL((blind_rotate_kernel_k2<LL, false>), dim3((unsigned)R), dim3(64), ldsk, s, a)
        BR_CASES(LAUNCH_K2)
#undef LAUNCH_K2
        HIP_TRY(c, hipGetLastError());
        name_kernel(c, "blind_rotate_kernel_k2<%d>", L);
        return TFHE_OK;
    }
    const int64_t tiny = c->br_tiny == -2 ? (int64_t)c->cu_count : c->br_tiny;
    if (tiny >= 0 && (int64_t)R <= tiny && (L == 2 || L == 3) && !c->br_rt_l) {      // (4 l waves per rotation: instantiated for the shipped l only)
        // every transform split over two waves: acc[2][N] | transposition buffers [4L][320] | extra slots [4L][256]
        H2Tables ht;
        ht.tw1h = c->d_tables + kH2TableOffset; ht.tw2q = ht.tw1h + 512; ht.tw3q = ht.tw2q + 64;
        // acc[2][N] | transposition buffers [4L][320] | extra slots [4L][256] | rotated differences [2 (step parity)][2][N]
        const size_t ldsh = 2 * kImg * 4 + (size_t)4 * L * (kH2Buf + 256) * sizeof(cplx) + 2 * 2 * kN * 4;
#define LAUNCH_H2_(LL, DG)                                                                                         \
        do {                                                                                                       \
            LDS_TRY(c, ldsh, blind_rotate_kernel_h2<LL, DG>); \
            hipLaunchKernelGGL((blind_rotate_kernel_h2<LL, DG>), dim3((unsigned)R), dim3(256 * LL), ldsh, s, a, ht); \
        } while (0)
#define LAUNCH_H2(LL) do { if (dg) LAUNCH_H2_(LL, true); else LAUNCH_H2_(LL, false); } while (0)
        BR_CASES(LAUNCH_H2)
#undef LAUNCH_H2
#undef LAUNCH_H2_
        HIP_TRY(c, hipGetLastError());
        name_kernel(c, "blind_rotate_kernel_h2<%d>", L);
        return TFHE_OK;
    }
    if (c->br_small >= 0 && (int64_t)R <= c->br_small) {
        // 27.4 KB of LDS and < 256 registers per wave: four workgroups per CU, 1024 rotations resident at two waves per SIMD
        const size_t ldsw = kW2LdsBytes;
        // two rotations per workgroup (lockstep through the step barrier, key reads shared in L1) when that fills the CUs evenly: from
        // more than one rotation up to one pair per CU (300 rotations 2.26 vs 2.58 ms, 512: 2.29 vs 2.57; 128-bit set 3.64 vs 4.10) and at (nearly) two
        // pairs per CU (1024: 3.35 vs 3.41; 128-bit 5.36 vs 5.57); in between single rotations spread better (700: 2.95 vs 3.29)
        const size_t cus2 = 2 * (size_t)c->cu_count;
        const bool pairs = !dg && (c->w2_rw == 2 || (c->w2_rw == 0 && ((R > cus2 / 2 && R <= cus2) || R > 2 * cus2 - cus2 / 8)));
#define LAUNCH_W2(LL)                                                                                              \
        if (dg) hipLaunchKernelGGL((blind_rotate_kernel_w2<LL, true>), dim3((unsigned)R), dim3(128), ldsw, s, a); \
        else if (pairs) hipLaunchKernelGGL((blind_rotate_kernel_w2<LL, false, 2>), dim3((unsigned)((R + 1) / 2)), dim3(256), 2 * ldsw, s, a); \
        else hipLaunchKernelGGL((blind_rotate_kernel_w2<LL, false>), dim3((unsigned)R), dim3(128), ldsw, s, a)
        BR_CASES_ANY_L(LAUNCH_W2)
#undef LAUNCH_W2
        HIP_TRY(c, hipGetLastError());
        if ((L == 2 || L == 3) && !c->br_rt_l) name_kernel(c, pairs ? "blind_rotate_kernel_w2<%d,rw2>" : "blind_rotate_kernel_w2<%d>", L);
        else name_kernel(c, pairs ? "blind_rotate_kernel_w2<0,rw2>(l=%d)" : "blind_rotate_kernel_w2<0>(l=%d)", L);
        return TFHE_OK;
    }
    {
        // one wave per rotation: half of a transform's key chunk requested a transform ahead, pass-B twiddles in registers
        // (round 3's <l, 16> / <l, 8, tw2 in LDS> variants were A/B scaffolding and are gone); four rotations per workgroup in
        // lockstep once the batch puts two waves on most SIMDs (option v3_rw: 0 = by batch size, 1, 4)
        const size_t lds3 = kV3LdsBytes;
        const bool group = c->v3_rw == 4 || (c->v3_rw == 0 && R >= 6 * (size_t)c->cu_count);      // 1536 on 256 CUs (1400 rotations: 5.47 vs 5.40 ms, 1700: 5.50 vs 5.66, 2000: 5.69 vs 5.93)
#define LAUNCH_V3_GROUP(LL, DG)                                                                                    \
        do {                                                                                                       \
            LDS_TRY(c, (4 * lds3), blind_rotate_kernel_v3<LL, 8, true, DG, 4>); \
            hipLaunchKernelGGL((blind_rotate_kernel_v3<LL, 8, true, DG, 4>), dim3((unsigned)((R + 3) / 4)), dim3(256), 4 * lds3, s, a); \
        } while (0)
#define LAUNCH_V3(LL)                                                                                              \
        if (group && dg) LAUNCH_V3_GROUP(LL, true);                                                                \
        else if (group) LAUNCH_V3_GROUP(LL, false);                                                                \
        else if (dg) hipLaunchKernelGGL((blind_rotate_kernel_v3<LL, 8, true, true>), dim3((unsigned)R), dim3(64), lds3, s, a);     \
        else hipLaunchKernelGGL((blind_rotate_kernel_v3<LL, 8, true, false>), dim3((unsigned)R), dim3(64), lds3, s, a)
        BR_CASES_ANY_L(LAUNCH_V3)
#undef LAUNCH_V3
#undef LAUNCH_V3_GROUP
        HIP_TRY(c, hipGetLastError());
        if ((L == 2 || L == 3) && !c->br_rt_l) name_kernel(c, group ? "blind_rotate_kernel_v3<%d,8,tw2reg,rw4>" : "blind_rotate_kernel_v3<%d,8,tw2reg>", L);
        else name_kernel(c, group ? "blind_rotate_kernel_v3<0,8,tw2reg,rw4>(l=%d)" : "blind_rotate_kernel_v3<0,8,tw2reg>(l=%d)", L);
        return TFHE_OK;
    }
}

// A batch whose size is not a multiple of what the chip holds pays for its last, partly filled round as for a full one
// when every rotation is one wave: the one-wave kernel (blind_rotate_kernel_v3) has 2048 rotations resident at two waves per
// SIMD, and a last round of r <= 1024 leaves most SIMDs with one wave or none for the 4 - 5 ms a rotation takes.  So the
// whole rounds go to the one-wave kernel and a last round of at most br_small (1024) rotations to the kernels that put two
// or 4 l waves on a rotation (blind_rotate_kernel_w2 / _h2), one launch after the other on the same stream.  Same device,
// interleaved (profiles/r04/r04c_split80.jsonl, 80-bit set; the parts alone: 2048 rotations 5.87 ms, 1024: 3.33, 512: 2.27):
//     2560 rotations 8.79 vs 9.84 ms in one launch, 3072: 9.53 vs 9.70, 5000: 15.29 vs 15.56, 6400: 19.58 vs 21.18;
//     128-bit set 3072: 15.43 vs 15.92, 5000: 24.57 vs 25.26.
// (Measured and removed: the same for a batch just above br_small — the first 1024 rotations on the two-wave kernel, up to one
//  rotation per CU more on the 4 l-wave kernel.  1100 rotations: 5.20 ms against 5.01 on the one-wave kernel (80-bit set), 8.41
//  against 8.41 (128-bit set): the second launch costs what the better packing saves.)
// (Measured dead end: the tail on a second stream, launched first so that the whole rounds move into the slots it frees —
//  3072 rotations 9.39 ms, but 2560: 10.7 and 5000: 15.8: whichever kernel the dispatcher favours starves the other.)
// Option br_split (default 1; 0: always one launch).
//
// k = 2 (blind_rotate_kernel_k2, lockstep groups of up to seven rotations, one workgroup per CU): a round of n rotations per
// CU takes (one device, 80-bit set with tlwe_mask_size 2, profiles/r04/r04w_k2_rounds.jsonl)
//     n = 1 .. 7:   6.8 / 6.8 / 7.2 / 7.3 / 9.4 / 9.5 / 10.3 ms
// — up to one wave per SIMD costs the same 7 ms, the second wave on a SIMD 2.1 - 3 ms more — so what pays is rounds of 6 - 7
// and a remainder of at most 4, not the equally full rounds of round 3 (4096 rotations = 16 per CU: 6 + 6 + 4 -> 26.4 ms by
// this table against 6 + 5 + 5 -> 28.4).  k2_partition() picks the round sizes by dynamic programming over that table; rounds
// of (nearly) equal size share a launch (the kernel deals a launch's rotations out in equally full rounds itself).
static const double kK2RoundCost[8] = {0.0, 6.8, 6.85, 7.2, 7.3, 9.45, 9.55, 10.3};
// ... and a LAST round of one or two rotations per CU on the three-waves-per-rotation kernel (blind_rotate_kernel_k2w3): 3.1 ms up
// to one rotation per CU (three waves on three SIMDs), 4.3 ms up to two (six waves on four SIMDs) — against 6.8 on the one-wave
// kernel (profiles/r05/r05f_k2_sweep.jsonl).  Same device, tlwe_parameters_80(tlwe_mask_size = 2): a single gate 3.11 vs 6.78 ms,
// 512 rotations 4.32 vs 6.84, 2048: 13.7 vs 14.5, 2304: 14.5 vs 16.8, 4096: 26.8 vs 27.3 (rounds of 7 + 7 per CU and 512
// rotations on the three-wave kernel instead of 6 + 6 + 4: the table says 24.9 — back-to-back launches cost 1 - 2 ms it does not know).
static const double kK2W3RoundCost[3] = {0.0, 3.1, 4.3};
struct K2Seg { size_t count; int kind; };       // kind: 0 = blind_rotate_kernel_k2 (lockstep groups), 1 = blind_rotate_kernel_k2w3
static std::vector<K2Seg> k2_partition(size_t R, size_t cus, bool allow_w3)
{
    const size_t q = (R + cus - 1) / cus;                    // rotations per CU, rounded up
    if (q <= 2 && allow_w3) return {{R, 1}};
    if (q <= 7 && !allow_w3) return {{R, 0}};
    std::vector<double> best(q + 1, 1e300);
    std::vector<int> take(q + 1, 0);
    best[0] = 0.0;
    for (size_t i = 1; i <= q; i++)
        for (int n = 1; n <= 7 && (size_t)n <= i; n++)
            if (best[i - n] + kK2RoundCost[n] < best[i]) { best[i] = best[i - n] + kK2RoundCost[n]; take[i] = n; }
    // the tail: nothing, or t = 1 / 2 rotations per CU on the three-wave kernel
    int tail = 0;
    double total = best[q];
    for (int t = 1; allow_w3 && t <= 2 && (size_t)t <= q; t++)
        if (best[q - t] + kK2W3RoundCost[t] < total) { total = best[q - t] + kK2W3RoundCost[t]; tail = t; }
    std::vector<int> rounds;
    for (size_t i = q - (size_t)tail; i > 0; i -= (size_t)take[i]) rounds.push_back(take[i]);
    std::sort(rounds.begin(), rounds.end(), std::greater<int>());
    // consecutive rounds of the same size -> one launch; the last launch takes what is left of R
    std::vector<K2Seg> seg;
    size_t done = 0;
    for (size_t i = 0; i < rounds.size() && done < R;) {
        size_t j = i;
        while (j < rounds.size() && rounds[j] == rounds[i]) j++;
        const size_t want = (size_t)rounds[i] * (j - i) * cus;
        if ((j == rounds.size() && !tail) || done + want >= R) { seg.push_back({R - done, 0}); done = R; break; }
        seg.push_back({want, 0});
        done += want;
        i = j;
    }
    if (done < R) seg.push_back({R - done, 1});
    return seg;
}

int32_t launch_blind_rotate(tfhe_ctx *c, size_t R, int32_t mu, hipStream_t s)
{
    DiagArgs diag;
    int32_t rc = prepare_diag(c, R, s, diag);
    if (rc) return rc;
    alloc_checkpoint();
    std::vector<K2Seg> seg;                                    // rotations per launch (and, k = 2, which kernel), in order
    const bool tuned_l = c->P.bs_l == 2 || c->P.bs_l == 3;
    if (c->anyn()) {
        // one launch
    } else if (c->br_split && !c->br_general && !c->measure_margin && c->P.N == kN && c->P.k == 2 && tuned_l && (c->k2_rw == 0 || c->k2_rw == 7) && c->k2_w3 != 1) {
        seg = k2_partition(R, (size_t)c->cu_count, c->k2_w3 < 0);
    } else if (c->br_split && !c->br_general && c->P.N == kN && c->P.k == 1 && c->br_small > 0) {      // (any l: the run-time-l instantiations)
        const size_t resident = 8 * (size_t)c->cu_count;      // rotations of blind_rotate_kernel_v3 on the chip
        if (R > resident && R % resident > 0 && R % resident <= (size_t)c->br_small) seg = {{R - R % resident, -1}, {R % resident, -1}};
    }
    if (seg.empty()) return launch_blind_rotate_part(c, 0, R, mu, s, diag);
    if (seg.size() == 1) return launch_blind_rotate_part(c, 0, R, mu, s, diag, seg[0].kind);
    std::string names;
    size_t first = 0;
    for (const K2Seg &sg : seg) {
        rc = launch_blind_rotate_part(c, first, sg.count, mu, s, diag, sg.kind);
        if (rc) return rc;
        if (names.empty() || names.substr(names.rfind(" + ") == std::string::npos ? 0 : names.rfind(" + ") + 3) != c->last_kernel)
            names += (names.empty() ? "" : " + ") + c->last_kernel;
        first += sg.count;
    }
    c->last_kernel = names;
    return TFHE_OK;
}

int32_t launch_keyswitch(tfhe_ctx *c, size_t G, const int32_t *e0, const int32_t *e1, const int32_t *dst,
                                const int32_t *ext, int32_t *out, hipStream_t s)
{
    KsArgs k;
    k.ext = ext;
    k.ks = c->d_ks;
    k.e0 = e0; k.e1 = e1; k.dst = dst;
    k.out = out;
    k.n = c->P.n; k.kN = c->P.k * c->P.N; k.t = c->P.ks_t; k.log2_base = c->P.ks_log2_base;
    const int n1 = c->P.n + 1;
    k.in_stride = k.kN + 1; k.in_off = 0; k.in_b = k.kN; k.out_stride = n1; k.out_off = 0; k.out_b = c->P.n; k.add_b = 1;
    if (c->ks_mode == 4) {
        Ks4Args a4;
        a4.ext = ext; a4.bmat = (const i32x4 *)c->d_ks4; a4.e0 = e0; a4.e1 = e1; a4.dst = dst; a4.out = out;
        a4.n = c->P.n; a4.kN = k.kN; a4.G = (int)G; a4.wtiles = c->ks4_wtiles;
        a4.in_stride = k.kN + 1; a4.in_off = 0; a4.in_b = k.kN; a4.out_stride = n1; a4.out_off = 0; a4.out_b = c->P.n; a4.add_b = 1;
        // split the mask words over several blocks, partial sums combined with exact integer atomics: 16 slices for
        // small batches (latency), 2 for large ones (two waves per SIMD so that one wave's MFMAs overlap the other's
        // A-fragment generation and LDS reads)
        a4.kslices = (k.kN % 512 != 0) ? 1 : (G <= 512 ? 16 : c->ks_slices_large);
        if (a4.kslices > 1) {
            Ks3Args i3;
            i3.ext = ext; i3.e0 = e0; i3.e1 = e1; i3.dst = dst; i3.out = out; i3.kN = k.kN; i3.n = c->P.n;
            i3.in_stride = a4.in_stride; i3.in_b = a4.in_b; i3.out_stride = n1; i3.out_b = c->P.n;
            hipLaunchKernelGGL(ks3_init_kernel, dim3((unsigned)G), dim3(256), 0, s, i3);
        }
        a4.Gpad = (int)((G + 63) / 64 * 64);
        HIP_TRY(c, c->abar.reserve((size_t)(k.kN / 4) * a4.Gpad * 16));
        a4.abar_t = (const i32x4 *)c->abar.p;
        hipLaunchKernelGGL(ks4_digits_kernel, dim3((unsigned)(a4.Gpad / 32), (unsigned)(k.kN / 128)), dim3(128), 0, s, a4, (i32x4 *)c->abar.p);
        hipLaunchKernelGGL(keyswitch_kernel_v4, dim3((unsigned)((G + 255) / 256), (unsigned)c->ks4_wtiles, (unsigned)a4.kslices), dim3(256), 0, s, a4);
        HIP_TRY(c, hipGetLastError());
        return TFHE_OK;
    }
    if (c->ks_mode == 3) {
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
    hipLaunchKernelGGL(keyswitch_kernel, dim3((unsigned)G, (unsigned)((n1 + 256 * KS1_WPT - 1) / (256 * KS1_WPT))), dim3(256), 0, s, k);     // any base, t, n
    HIP_TRY(c, hipGetLastError());
    return TFHE_OK;
}

// mk_keyswitch (mk_internals.jl:397-411): per party a single-key keyswitch of its mask column with b = 0, the b words chained
// through the output (stream-ordered).  ext = the context's extracted samples [B][P N + 1], out: [B][P n + 1].
int32_t launch_mk_keyswitch(tfhe_ctx *c, size_t B, const int32_t *d_gate, int32_t *out, hipStream_t s)
{
    const int NP = c->mk_parties, n = c->P.n, nw = NP * n + 1, Nn = c->P.N, ew = NP * Nn + 1;
    if (c->ks_mode == 4) {
        Ks4Args a4;
        a4.ext = (const int32_t *)c->ext.p; a4.e0 = d_gate; a4.e1 = nullptr; a4.dst = nullptr; a4.out = out;
        a4.n = n; a4.kN = Nn; a4.G = (int)B; a4.wtiles = c->ks4_wtiles;
        a4.in_stride = ew; a4.in_b = NP * Nn; a4.out_stride = nw; a4.out_b = NP * n;
        for (int p = 0; p < NP; p++) {
            a4.in_off = p * Nn; a4.out_off = p * n; a4.add_b = (p == 0); a4.kslices = 1;
            a4.Gpad = (int)((B + 63) / 64 * 64);
            HIP_TRY(c, c->abar.reserve((size_t)(Nn / 4) * a4.Gpad * 16));
            a4.abar_t = (const i32x4 *)c->abar.p;
            hipLaunchKernelGGL(ks4_digits_kernel, dim3((unsigned)(a4.Gpad / 32), (unsigned)(Nn / 128)), dim3(128), 0, s, a4, (i32x4 *)c->abar.p);
            a4.bmat = (const i32x4 *)c->d_mk_ks4 + (size_t)p * c->mk_ks4_frags;
            hipLaunchKernelGGL(keyswitch_kernel_v4, dim3((unsigned)((B + 255) / 256), (unsigned)c->ks4_wtiles), dim3(256), 0, s, a4);
        }
    } else if (c->ks_mode == 1) {
        // any base / length: the gather kernel per party
        KsArgs k1;
        k1.ext = (const int32_t *)c->ext.p; k1.e0 = d_gate; k1.e1 = nullptr; k1.dst = nullptr; k1.out = out;
        k1.n = n; k1.kN = c->P.N; k1.t = c->P.ks_t; k1.log2_base = c->P.ks_log2_base;
        k1.in_stride = ew; k1.in_b = NP * c->P.N; k1.out_stride = nw; k1.out_b = NP * n;
        for (int p = 0; p < NP; p++) {
            k1.in_off = p * c->P.N; k1.out_off = p * n; k1.add_b = (p == 0);
            k1.ks = c->d_ks + (size_t)p * c->mk_ksp_words;
            hipLaunchKernelGGL(keyswitch_kernel, dim3((unsigned)B, (unsigned)((n + 1 + 256 * KS1_WPT - 1) / (256 * KS1_WPT))), dim3(256), 0, s, k1);
        }
    } else {
        Ks3Args k3;
        k3.ext = (const int32_t *)c->ext.p; k3.e0 = d_gate; k3.e1 = nullptr; k3.dst = nullptr; k3.out = out;
        k3.n = n; k3.kN = Nn; k3.t = c->P.ks_t; k3.log2_base = 2; k3.stride = c->ks_stride; k3.G = (int)B;
        k3.in_stride = ew; k3.in_b = NP * Nn; k3.out_stride = nw; k3.out_b = NP * n;
        k3.in_off = 0; k3.out_off = 0; k3.ksp = c->d_mk_ksp;
        hipLaunchKernelGGL(ks3_init_kernel, dim3((unsigned)B), dim3(256), 0, s, k3);
        const unsigned tiles = (unsigned)((B + KS3_G - 1) / KS3_G);
        for (int p = 0; p < NP; p++) {
            k3.in_off = p * Nn; k3.out_off = p * n; k3.ksp = c->d_mk_ksp + (size_t)p * c->mk_ksp_words;
            hipLaunchKernelGGL(keyswitch_kernel_v3, dim3(tiles * KS3_SLICES, (unsigned)((c->ks_stride + 511) / 512)), dim3(128), 0, s, k3);
        }
    }
    HIP_TRY(c, hipGetLastError());
    return TFHE_OK;
}
