// tfhe_engine.hip — MI355X (gfx950) TFHE gate-bootstrapping engine: context, key loading, launchers, C ABI
// (include/tfhe_mi355x.h).  The kernels live in the three headers included below; this file is the only
// translation unit.
//
// Pipeline of one batch call (tfhe_gates_batch* / tfhe_gates_level):
//   prologue_kernel          gate affine prologue (gates.jl) + modulus switch (bootstrap.jl:74-75)
//   blind_rotate_kernel_*    accumulator resident in LDS for all n CMUX steps (bootstrap.jl:19-59, tgsw.jl:99-129,
//                            polynomials.jl:106-132), fused test-vector init and sample extraction (tlwe.jl:55-59);
//                            variant chosen by parameters and batch size (launch_blind_rotate)
//   ks4_digits_kernel +      keyswitch (keyswitch.jl:45-80) as an exact int8 MFMA contraction, MUX add fused
//   keyswitch_kernel_v4      (fallbacks: keyswitch_kernel_v3 / keyswitch_kernel)
//   trivial_gates_kernel     NOT / CONSTANT / COPY (gates.jl:76-93)
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

#include "kernels_gates.hpp"
#include "kernels_blind_rotate.hpp"
#include "kernels_keyswitch.hpp"

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
    hipStream_t last_stream = nullptr;   // stream of the previous batch call: the workspaces are shared, so a call on a
                                         // different stream first waits for the previous stream (enter_stream)
    bool timing_valid = false;
    int64_t last_rotations = 0;
    int ks_slices_large = 2;     // K-split of the MFMA keyswitch for large batches (tfhe_set_option("ks_slices", 1|2|4))
    int ks_variant = 4;          // 1 = one workgroup per sample, 3 = tiled + sliced + XCD-aware integer VALU, 4 = int8 MFMA (default)
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
    DevBuf bara, ext, map, io[4], margin, abar;
    size_t margin_rows = 0;
    bool mk_force_general = false; // tfhe_set_option("mk_general", 1): use the any-P kernel for 2 parties too (cross-check)
    bool measure_margin = false;   // tfhe_set_option("measure_margin", 1): blind rotations also record their rounding margin
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
    if (p.bs_l > (p.parties > 1 ? 8 : 4))
        return fail(TFHE_ERR_UNSUPPORTED, "tfhe_ctx_create: bs_decomp_length > 4 (single key) / > 8 (multi-key) unsupported");
    if (p.parties > 8) return fail(TFHE_ERR_UNSUPPORTED, "tfhe_ctx_create: more than 8 parties unsupported");
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
    c->bara.release(); c->ext.release(); c->map.release(); c->margin.release(); c->abar.release();
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
    c->margin_rows = 0;
    BrArgs a;
    a.margin = nullptr;
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
    if ((c->br_small >= 0 && (int64_t)R <= c->br_small) && c->br_variant >= 2 && !c->measure_margin) {
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
    if (c->measure_margin && c->P.k == 1 && c->P.N == kN) {   // diagnostics: same kernel, MARGIN instantiation
        const size_t ldsm = 2 * kN * 4 + (kXchElems + 64) * sizeof(cplx);
        HIP_TRY(c, c->margin.reserve(R * sizeof(double)));
        a.margin = (double *)c->margin.p;
        switch (c->P.bs_l) {
        case 1: hipLaunchKernelGGL((blind_rotate_kernel_v3<1, 8, false, true>), dim3((unsigned)R), dim3(64), ldsm, s, a); break;
        case 2: hipLaunchKernelGGL((blind_rotate_kernel_v3<2, 8, false, true>), dim3((unsigned)R), dim3(64), ldsm, s, a); break;
        case 3: hipLaunchKernelGGL((blind_rotate_kernel_v3<3, 8, false, true>), dim3((unsigned)R), dim3(64), ldsm, s, a); break;
        case 4: hipLaunchKernelGGL((blind_rotate_kernel_v3<4, 8, false, true>), dim3((unsigned)R), dim3(64), ldsm, s, a); break;
        default: return c->set_err(TFHE_ERR_UNSUPPORTED, "blind rotate: bs_l = %d unsupported", c->P.bs_l);
        }
        HIP_TRY(c, hipGetLastError());
        c->margin_rows = R;
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
    if (c->ks_variant == 4 && c->d_ks4 && k.kN % 128 == 0) {
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
    if (n1 <= 256) hipLaunchKernelGGL((keyswitch_kernel<1>), dim3((unsigned)G), dim3(256), 0, s, k);
    else if (n1 <= 512) hipLaunchKernelGGL((keyswitch_kernel<2>), dim3((unsigned)G), dim3(256), 0, s, k);
    else hipLaunchKernelGGL((keyswitch_kernel<4>), dim3((unsigned)G), dim3(256), 0, s, k);
    HIP_TRY(c, hipGetLastError());
    return TFHE_OK;
}

// Workspaces (bara, ext, abar, map, margin) are per context: work enqueued on another stream must have finished
// before a call on stream `s` reuses them.
static int32_t enter_stream(tfhe_ctx *c, hipStream_t s)
{
    if (c->last_stream && c->last_stream != s) HIP_TRY(c, hipStreamSynchronize(c->last_stream));
    c->last_stream = s;
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
    {
        const int32_t rc0 = enter_stream(c, s);
        if (rc0) return rc0;
    }
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
    { const int32_t rc0 = enter_stream(c, s); if (rc0) return rc0; }
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
    { const int32_t rc0 = enter_stream(c, s); if (rc0) return rc0; }
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
    if (parties < 2 || parties > 8 || c->P.parties < parties)
        return c->set_err(TFHE_ERR_INVALID_ARG, "mk_load_bootstrap_key: parties must be 2..8 and not exceed the context's max_parties (mk_api.jl:94)");
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
    if (parties < 2 || parties > 8 || c->P.parties < parties)
        return c->set_err(TFHE_ERR_INVALID_ARG, "mk_load_keyswitch_key: parties must be 2..8 and not exceed the context's max_parties");
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
    { const int32_t rc0 = enter_stream(c, s); if (rc0) return rc0; }
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
    const size_t lds = (size_t)(NP + 1) * kN * 4 + (kXchElems + 64) * sizeof(cplx);
    const bool special = (NP == 2 && c->P.bs_l >= 2 && c->P.bs_l <= 4 && !c->mk_force_general);
    if (special) {
        switch (c->P.bs_l) {
        case 2: hipLaunchKernelGGL((mk_blind_rotate_kernel<2>), dim3((unsigned)B), dim3(64), lds, s, a); break;
        case 3: hipLaunchKernelGGL((mk_blind_rotate_kernel<3>), dim3((unsigned)B), dim3(64), lds, s, a); break;
        default: hipLaunchKernelGGL((mk_blind_rotate_kernel<4>), dim3((unsigned)B), dim3(64), lds, s, a); break;
        }
    } else {
        MkGenArgs ga;
        ga.bara = a.bara; ga.bk = a.bk; ga.ext = a.ext; ga.T = a.T; ga.g = a.g; ga.n = n; ga.mu = a.mu; ga.parties = NP; ga.L = c->P.bs_l;
        if (lds > 64 * 1024)
            HIP_TRY(c, hipFuncSetAttribute((const void *)mk_blind_rotate_kernel_general, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(mk_blind_rotate_kernel_general, dim3((unsigned)B), dim3(64), lds, s, ga);
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
            a4.Gpad = (int)((B + 63) / 64 * 64);
            HIP_TRY(c, c->abar.reserve((size_t)(kN / 4) * a4.Gpad * 16));
            a4.abar_t = (const i32x4 *)c->abar.p;
            hipLaunchKernelGGL(ks4_digits_kernel, dim3((unsigned)(a4.Gpad / 32), (unsigned)(kN / 128)), dim3(128), 0, s, a4, (i32x4 *)c->abar.p);
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

int32_t tfhe_last_rounding_margin(tfhe_ctx *c, double *worst)
{
    if (!c || !worst) return TFHE_ERR_INVALID_ARG;
    if (!c->margin_rows) return c->set_err(TFHE_ERR_STATE, "last_rounding_margin: enable tfhe_set_option(\"measure_margin\", 1) before the batch call (N = 1024, k = 1)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    std::vector<double> h(c->margin_rows);
    HIP_TRY(c, hipMemcpy(h.data(), c->margin.p, c->margin_rows * sizeof(double), hipMemcpyDeviceToHost));
    double m = 0;
    for (double v : h) m = v > m ? v : m;
    *worst = m;
    return TFHE_OK;
}

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
    if (!strcmp(name, "ks_slices")) {
        if (value != 1 && value != 2 && value != 4) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: ks_slices must be 1, 2 or 4");
        c->ks_slices_large = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "measure_margin")) { c->measure_margin = value != 0; return TFHE_OK; }
    if (!strcmp(name, "mk_general")) { c->mk_force_general = value != 0; return TFHE_OK; }
    if (!strcmp(name, "ks_variant")) {
        if (value != 1 && value != 3 && value != 4) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: ks_variant must be 1, 3 or 4");
        c->ks_variant = (int)value;
        return TFHE_OK;
    }
    return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: unknown option '%s'", name);
}

}  // extern "C"
