// engine_context.hip — context: creation / destruction, twiddle tables, the sharding rule, page-locked host memory, last error
#include "engine.hpp"
#include <cmath>

thread_local ErrBuf g_create_error;
thread_local const tfhe_ctx *g_rejected_ctx = nullptr;
// a failure reported to THIS thread about g_caught_ctx without touching that context's own buffer (abi_caught)
static thread_local const tfhe_ctx *g_caught_ctx = nullptr;
static thread_local ErrBuf g_caught_error;
std::atomic<int64_t> g_fail_alloc_countdown{0};
const char kBusyMessage[] = "the context is inside another call on another thread: calls on one context must not overlap (one caller at a time; use one context per thread)";

int32_t abi_caught(tfhe_ctx *c, const char *who) noexcept
{
    int32_t code = TFHE_ERR_STATE;
    ErrBuf &e = c ? g_caught_error : g_create_error;
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = TFHE_ERR_NOMEM;
        e.format("%s: out of host memory (std::bad_alloc); nothing was changed that a repeated call would not redo", who);
    } catch (const std::exception &x) {
        e.format("%s: unexpected C++ exception: %s", who, x.what());
    } catch (...) {
        e.format("%s: unexpected C++ exception", who);
    }
    g_caught_ctx = c;
    if (c) {
        const tfhe_ctx *rejected = g_rejected_ctx;
        CallGuard again(c);
        if (again.ok) c->err = g_caught_error;
        g_rejected_ctx = rejected;
    }
    return code;
}

static void build_tables(std::vector<cplx> &h)
{
    h.resize(kN512TableOffset + kN512TableElems);
    fill_tables<long double>(h.data(), [](long double a) { return cosl(a); }, [](long double a) { return sinl(a); });
    // N = 2048: tw1f2[w][q][t] = e^{-i pi t (1 + 4w + 8q) / 2048}
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int w = 0; w < 2; w++)
        for (int q = 0; q < 8; q++)
            for (int t = 0; t < 64; t++) {
                const long double a = -pi * (long double)(t * (1 + 4 * w + 8 * q)) / 2048.0L;
                h[kTableElems + w * 512 + q * 64 + t] = mk((double)cosl(a), (double)sinl(a));
            }
    // split 256-point transforms (blind_rotate_kernel_h2): angles in units of pi
    auto put = [&](size_t idx, long double turns_of_pi) { const long double a = -pi * turns_of_pi; h[idx] = mk((double)cosl(a), (double)sinl(a)); };
    for (int hh = 0; hh < 2; hh++)
        for (int q = 0; q < 4; q++)
            for (int t = 0; t < 64; t++)      // e^{-i pi t/N} * (h ? e^{-2 pi i t/512} : 1) * e^{-2 pi i t q/256}
                put(kH2TableOffset + (size_t)(hh * 4 + q) * 64 + t, (long double)t / 1024.0L + (hh ? (long double)t / 256.0L : 0.0L) + (long double)(t * q) / 128.0L);
    for (int q = 0; q < 4; q++)
        for (int t = 0; t < 16; t++) put(kH2TableOffset + 512 + (size_t)q * 16 + t, (long double)(t * q) / 32.0L);      // e^{-2 pi i t q/64}
    for (int q = 0; q < 4; q++)
        for (int t = 0; t < 4; t++) put(kH2TableOffset + 576 + (size_t)q * 4 + t, (long double)(t * q) / 8.0L);          // e^{-2 pi i t q/16}
    // N = 512 (blind_rotate_kernel_n512): first-pass twiddles with the lane part of that degree's twist: e^{-i pi t/512} e^{-2 pi i t q/256}
    for (int q = 0; q < 4; q++)
        for (int t = 0; t < 64; t++) put(kN512TableOffset + (size_t)q * 64 + t, (long double)t / 512.0L + (long double)(t * q) / 128.0L);
}

// ---- sharding of a gate stream: contiguous shards balanced by blind-rotation count ----------
// MUX = 2 rotations, NOT / CONST / COPY = 0 (gates.jl:163-177, 76-93); every gate also weighs 1/1000 so that cut
// points stay well defined among trivial gates.  bounds[r] .. bounds[r+1] is shard r.  Same rule as
// tfhe.jl_amd/sharding.py:shard_bounds (tests compare the two).
void shard_bounds_by_rotations(const uint8_t *opcodes, int64_t B, int shards, int64_t *bounds)
{
    auto cost = [&](int64_t g) -> int64_t {
        const int op = opcodes ? opcodes[g] : TFHE_GATE_NAND;
        const int rot = op == TFHE_GATE_MUX ? 2 : (op == TFHE_GATE_NOT || op == TFHE_GATE_COPY || op == TFHE_GATE_CONST0 || op == TFHE_GATE_CONST1) ? 0 : 1;
        return 1000 * (int64_t)rot + 1;
    };
    int64_t total = 0;
    for (int64_t g = 0; g < B; g++) total += cost(g);
    bounds[0] = 0;
    int64_t cum = 0, g = 0;
    for (int r = 1; r < shards; r++) {
        // smallest g with cum(g) * shards >= total * r  (cum(g) = cost of gates [0, g))
        while (g < B && cum * shards < total * r) { cum += cost(g); g++; }
        bounds[r] = g;
    }
    bounds[shards] = B;
}

extern "C" {

int32_t tfhe_abi_version(void) { return TFHE_ABI_VERSION_REPORTED; }      // (negative: a development build, experiment.hpp)

int32_t tfhe_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return -1;
    return n;
}

const char *tfhe_last_error(const tfhe_ctx *ctx)
{
    if (!ctx) return g_create_error.c_str();
    if (g_rejected_ctx == ctx) return kBusyMessage;      // this thread's last call on ctx was refused because another thread was inside one
    if (g_caught_ctx == ctx) return g_caught_error.c_str();      // ... or ended in a C++ exception (ABI_CATCH)
    return ctx->err.c_str();
}

int32_t tfhe_shard_bounds(const uint8_t *opcodes, int64_t B, int32_t shards, int64_t *bounds) try
{
    alloc_checkpoint();
    if (B < 0 || shards < 1 || !bounds) return TFHE_ERR_INVALID_ARG;
    shard_bounds_by_rotations(opcodes, B, shards, bounds);
    return TFHE_OK;
}
ABI_CATCH(nullptr, "tfhe_shard_bounds")

// ---- is this parameter set inside what a Float64 transform computes exactly? --------------------------------------------
// The external product (tgsw.jl:125-129) sums `np` negacyclic products digit (*) key per output polynomial — np = (k + 1) l
// single-key, (P + 1) l multi-key (mk_internals.jl:368-387, summed in the spectrum domain here) — through a Float64 transform and
// ONE rounding per coefficient (polynomials.jl:115-116).  Two things must hold for the rounded word to be the exact product:
//   (i)  |pre-rounding value| < 2^51: the kernels round with the 1.5 * 2^52 trick (br_core.hpp round_to_torus32).
//        Worst case over ANY Int32 key words: np N 2^(beta-1) 2^31.  With a real key (uniform words, digits uniform in
//        [-Bg/2, Bg/2)) the values are sums of np N independent terms: rms = sqrt(np N) (2^beta / sqrt 12) (2^32 / sqrt 12).
//   (ii) the transform's rounding error stays below 1/2.  It grows like eps x rms x max(log2(N/2), 4); every DIAG measurement of
//        rounds 2-6 (the shipped kernels and 13 000 fuzzed sets: N = 2 .. 8192, l = 1 .. 8, beta = 1 .. 12, k = 1 .. 6, 2 .. 9
//        parties) lies between 1.4 and 4.01 times eps x rms x max(log2(N/2), 4) with eps = 2^-53; the prediction uses 4.5.
// exact_domain = 2: (i) holds for every key whatever its words and the predicted margin is below 1/4;
//                1: (i) holds for real keys (8 rms < 2^51) and the predicted margin is below 1/4 — tfhe_parameters_80 is here: its
//                   all-keys bound is exactly 2^52 (header, "Exactness domain"), one bit above, like the reference's own;
//                0: outside — the engine computes, as the reference does (polynomials.jl:135-144 warns there too), but a word may
//                   differ from the exact product; tfhe_last_rounding_margin tells.
static void exactness_class(const tfhe_params &p, int &cls, double &bound_log2, double &margin)
{
    const double np = (double)(p.parties > 1 ? p.parties + 1 : p.k + 1) * p.bs_l, N = (double)p.N, beta = (double)p.bs_log2_base;
    bound_log2 = std::log2(np * N) + (beta - 1.0) + 31.0;
    const double rms = std::sqrt(np * N) * std::exp2(beta + 32.0) / 12.0;
    margin = 4.5 * std::exp2(-53.0) * rms * std::max(4.0, std::log2(N / 2.0));
    const bool margin_ok = margin < 0.25, typical_ok = 8.0 * rms < std::exp2(51.0);
    cls = !(margin_ok && typical_ok) ? 0 : bound_log2 < 51.0 ? 2 : 1;
}

int32_t tfhe_ctx_create(const tfhe_params *params, int32_t device_id, tfhe_ctx **out_ctx) try
{
    alloc_checkpoint();
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
    // Every parameter set the reference would accept is accepted (SchemeParameters is an unvalidated positional struct and
    // tlwe_mask_size a free keyword, api.jl:4-21,30,55; the transform works for any even length, polynomials.jl:44-58): tuned
    // kernels where one was instantiated (launch_blind_rotate_part), blind_rotate_kernel_general for the other shapes at
    // N = 1024 / 2048 with k <= 4, the any-N kernels (kernels_anyn.hpp) for everything else.  What remains refused:
    //   * N > 8192: one polynomial's transform buffer and digit words no longer fit the 160 KB of LDS of a CU (and the
    //     Float64 transform of the reference itself has lost its exactness long before: polynomials.jl:115-116);
    //   * multi-key with tlwe_mask_size != 1: the reference's multi-key code hard-wires k = 1 (mk_internals.jl:89-91,129-131).
    if (p.N > 8192) {
        snprintf(buf, sizeof buf, "tfhe_ctx_create: N = %d > 8192 unsupported (one polynomial's transform no longer fits a CU's LDS)", p.N);
        return fail(TFHE_ERR_UNSUPPORTED, buf);
    }
    if (p.k != 1 && p.parties != 1) return fail(TFHE_ERR_UNSUPPORTED, "tfhe_ctx_create: multi-key needs tlwe_mask_size 1 (as the reference, mk_internals.jl:89-91)");
    if (p.n > (1 << 24) || p.k > 1024 || p.parties > 1024) return fail(TFHE_ERR_INVALID_ARG, "tfhe_ctx_create: lwe_size, tlwe_mask_size or max_parties beyond any plausible value");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        snprintf(buf, sizeof buf, "tfhe_ctx_create: no HIP device available (%s)", hipGetErrorString(e));
        return fail(TFHE_ERR_DEVICE, buf);
    }
    if (device_id < 0 || device_id >= ndev) return fail(TFHE_ERR_INVALID_ARG, "tfhe_ctx_create: device_id out of range");

    tfhe_ctx *c = new (std::nothrow) tfhe_ctx();
    if (!c) return fail(TFHE_ERR_NOMEM, "tfhe_ctx_create: out of host memory");
    c->P = p;
    c->device = device_id;
    c->g = make_gadget(p.bs_l, p.bs_log2_base);
    exactness_class(p, c->exact_domain, c->exact_bound_log2, c->exact_margin);
    // (from here on `c` is owned by this function until *out_ctx takes it: an exception — the twiddle vectors below — must not leak it)
    struct Owner { tfhe_ctx *c; ~Owner() { if (c) tfhe_ctx_destroy(c); } } owner{c};
    auto bail = [&](hipError_t err, const char *what) {
        snprintf(buf, sizeof buf, "tfhe_ctx_create: %s failed: %s", what, hipGetErrorString(err));
        g_create_error = buf;
        return (int32_t)TFHE_ERR_DEVICE;      // (Owner destroys c)
    };
    if ((e = hipSetDevice(device_id)) != hipSuccess) return bail(e, "hipSetDevice");
    {
        int cus = 0;
        if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id)) != hipSuccess) return bail(e, "hipDeviceGetAttribute");
        c->cu_count = cus > 0 ? cus : 256;
        // the batch-size thresholds of the dispatcher are counts of rotations PER CU measured on a 256-CU MI355X; they are kept in
        // those units so that a partitioned device (CPX: 32 CUs per partition) or another part of the family switches kernels at
        // the same fill levels: 4 per CU = what the chip holds of the two-wave kernel at two waves per SIMD (1024 on 256 CUs),
        // 16 per CU = two rounds of the one-wave kernel (4096).  tfhe_set_option overrides them with absolute counts.
        c->br_small = 4 * (int64_t)c->cu_count;
        c->pipeline_min = 16 * (int64_t)c->cu_count;
        c->level_split_min = 16 * (int64_t)c->cu_count;
    }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    for (auto &set : c->evring)
        for (auto &ev : set)
            if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
    for (auto &st : c->map_stage)
        if ((e = hipEventCreateWithFlags(&st.ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    c->map_ev = c->map_stage[0].ev;
    if ((e = hipEventCreateWithFlags(&c->done_ev, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    std::vector<cplx> h;
    build_tables(h);
    if ((e = hipMalloc((void **)&c->d_tables, h.size() * sizeof(cplx))) != hipSuccess) return bail(e, "hipMalloc(tables)");
    if ((e = hipMemcpyAsync(c->d_tables, h.data(), h.size() * sizeof(cplx), hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return bail(e, "hipMemcpy(tables)");
    c->T = tables_from(c->d_tables);
    {   // the any-N kernels' tables for this context's N: e^{-2 pi i t/M} | e^{-i pi j/N}, M = N/2 (long double on the host)
        const int M = p.N / 2;
        std::vector<cplx> ht((size_t)2 * M);
        const long double pi = 3.14159265358979323846264338327950288L;
        for (int t = 0; t < M; t++) {
            const long double a = -2.0L * pi * (long double)t / (long double)M, b = -pi * (long double)t / (long double)p.N;
            ht[(size_t)t] = mk((double)cosl(a), (double)sinl(a));
            ht[(size_t)M + t] = mk((double)cosl(b), (double)sinl(b));
        }
        if ((e = hipMalloc((void **)&c->d_anyn_tab, ht.size() * sizeof(cplx))) != hipSuccess) return bail(e, "hipMalloc(any-N tables)");
        if ((e = hipMemcpyAsync(c->d_anyn_tab, ht.data(), ht.size() * sizeof(cplx), hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
            (e = hipStreamSynchronize(c->stream)) != hipSuccess)
            return bail(e, "hipMemcpy(any-N tables)");
    }
    owner.c = nullptr;
    *out_ctx = c;
    return TFHE_OK;
}
ABI_CATCH(nullptr, "tfhe_ctx_create")

int32_t tfhe_ctx_create_multi(const tfhe_params *params, const int32_t *device_ids, int32_t n_dev, tfhe_ctx **out_ctx) try
{
    alloc_checkpoint();
    if (!params || !out_ctx || !device_ids) { g_create_error = "tfhe_ctx_create_multi: NULL argument"; return TFHE_ERR_INVALID_ARG; }
    *out_ctx = nullptr;
    if (n_dev < 1 || n_dev > 64) { g_create_error = "tfhe_ctx_create_multi: n_dev must be 1..64"; return TFHE_ERR_INVALID_ARG; }
    tfhe_ctx *c = new (std::nothrow) tfhe_ctx();
    if (!c) { g_create_error = "tfhe_ctx_create_multi: out of host memory"; return TFHE_ERR_NOMEM; }
    struct Owner { tfhe_ctx *c; ~Owner() { if (c) tfhe_ctx_destroy(c); } } owner{c};
    c->P = *params;
    c->device = device_ids[0];
    exactness_class(c->P, c->exact_domain, c->exact_bound_log2, c->exact_margin);
    c->kids.reserve((size_t)n_dev);
    for (int i = 0; i < n_dev; i++) {
        tfhe_ctx *k = nullptr;
        const int32_t rc = tfhe_ctx_create(params, device_ids[i], &k);
        if (rc) return rc;      // g_create_error holds the reason (Owner destroys c and the kids made so far)
        c->kids.push_back(k);
    }
    c->kid_ran.assign((size_t)n_dev, 0);
    c->cu_count = c->kids[0]->cu_count;
    c->level_split_min = c->kids[0]->level_split_min;
    // Device-to-device copies between the replicas of the wire table (pull_wires): allowed between two kids on the same
    // device and wherever hipDeviceCanAccessPeer says so; peer access is switched on for those pairs here, once.  Pairs
    // without it exchange rows through pinned host memory instead.
    c->peer_ok.assign((size_t)n_dev * n_dev, 0);
    c->xfer.assign((size_t)n_dev * n_dev, nullptr);
    for (int a = 0; a < n_dev; a++)
        for (int b = 0; b < n_dev; b++) {
            const int da = device_ids[a], db = device_ids[b];
            if (da == db) { c->peer_ok[(size_t)a * n_dev + b] = 1; continue; }
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, da, db) != hipSuccess || !can) { (void)hipGetLastError(); continue; }
            if (hipSetDevice(da) != hipSuccess) { (void)hipGetLastError(); continue; }
            const hipError_t ep = hipDeviceEnablePeerAccess(db, 0);
            if (ep == hipSuccess || ep == hipErrorPeerAccessAlreadyEnabled) c->peer_ok[(size_t)a * n_dev + b] = 1;
            (void)hipGetLastError();
        }
    // a device-to-device copy is issued on the DESTINATION's stream and reads the SOURCE's buffer (pull_wires): it is taken only
    // between pairs that may access each other in both directions
    for (int a = 0; a < n_dev; a++)
        for (int b = a + 1; b < n_dev; b++) {
            const uint8_t both = c->peer_ok[(size_t)a * n_dev + b] && c->peer_ok[(size_t)b * n_dev + a];
            c->peer_ok[(size_t)a * n_dev + b] = c->peer_ok[(size_t)b * n_dev + a] = both;
        }
    owner.c = nullptr;
    *out_ctx = c;
    return TFHE_OK;
}
ABI_CATCH(nullptr, "tfhe_ctx_create_multi")

int32_t tfhe_ctx_device_count(const tfhe_ctx *ctx) { return !ctx ? -1 : ctx->multi() ? (int32_t)ctx->kids.size() : 1; }

void tfhe_ctx_destroy(tfhe_ctx *c)
{
    if (!c) return;
    if (g_caught_ctx == c) g_caught_ctx = nullptr;
    if (g_rejected_ctx == c) g_rejected_ctx = nullptr;
    if (c->multi()) {
        for (tfhe_ctx *k : c->kids)        // transfers between the replicas still in flight use the buffers freed below
            if (k && k->stream) { (void)hipSetDevice(k->device); (void)hipStreamSynchronize(k->stream); }
        for (PairXfer *x : c->xfer) {
            if (!x) continue;
            for (auto &sl : x->slot) {
                sl.out.release(); sl.idx_src.release(); sl.in.release(); sl.idx_dst.release();
                if (sl.h_idx) (void)hipHostFree(sl.h_idx);
                if (sl.h_rows) (void)hipHostFree(sl.h_rows);
                if (sl.ready) (void)hipEventDestroy(sl.ready);
                if (sl.taken) (void)hipEventDestroy(sl.taken);
            }
            delete x;
        }
        for (tfhe_ctx *k : c->kids) tfhe_ctx_destroy(k);
        delete c;
        return;
    }
    if (c->twin) { tfhe_ctx_destroy(c->twin); c->twin = nullptr; }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->done_pending && c->done_ev) (void)hipEventSynchronize(c->done_ev);      // a call still running on a caller's stream
    if (c->d_tables) (void)hipFree(c->d_tables);
    if (c->d_anyn_tab) (void)hipFree(c->d_anyn_tab);
    if (c->borrows_keys) c->d_bk = nullptr, c->d_ks = nullptr, c->d_ksp = nullptr, c->d_ks4 = nullptr;      // the owner frees them
    if (c->d_bk) (void)hipFree(c->d_bk);
    if (c->d_ks) (void)hipFree(c->d_ks);
    if (c->d_ksp) (void)hipFree(c->d_ksp);
    if (c->d_ks4) (void)hipFree(c->d_ks4);
    if (c->d_wires) (void)hipFree(c->d_wires);
    if (c->d_mk_ks4) (void)hipFree(c->d_mk_ks4);
    if (c->d_mk_bk) (void)hipFree(c->d_mk_bk);
    if (c->d_mk_ksp) (void)hipFree(c->d_mk_ksp);
    c->bara.release(); c->ext.release(); c->map.release(); c->diag.release(); c->abar.release(); c->mk_acc.release(); c->spec.release();
    for (auto &b : c->io) b.release();
    for (auto &st : c->map_stage) {
        if (st.h) (void)hipHostFree(st.h);
        if (st.ev) (void)hipEventDestroy(st.ev);
    }
    for (auto &set : c->evring)
        for (auto &ev : set) if (ev) (void)hipEventDestroy(ev);
    if (c->done_ev) (void)hipEventDestroy(c->done_ev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int32_t tfhe_host_alloc(size_t bytes, void **out_ptr) try
{
    alloc_checkpoint();
    if (!out_ptr) return TFHE_ERR_INVALID_ARG;
    *out_ptr = nullptr;
    if (bytes == 0) return TFHE_OK;
    const hipError_t e = hipHostMalloc(out_ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { g_create_error.format("tfhe_host_alloc: %s", hipGetErrorString(e)); *out_ptr = nullptr; return TFHE_ERR_DEVICE; }
    return TFHE_OK;
}
ABI_CATCH(nullptr, "tfhe_host_alloc")

void tfhe_host_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }

// Blocks until everything queued on the context so far — on its own stream, its second stream, every device of a multi-device
// context — has completed.  The ONE entry point that may be called from any thread while another thread is inside a call on the
// same context (it takes no CallGuard): it reads stream handles that are fixed for the context's life and touches no workspace
// and no bookkeeping.  For whoever must free host buffers of a submitted batch without owning the context: a garbage
// collector's finalizer (julia/TFHEMI355X abandon!), an error path.
int32_t tfhe_ctx_synchronize(tfhe_ctx *c)
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    int32_t rc = TFHE_OK;
    if (c->multi()) {
        for (tfhe_ctx *k : c->kids) { const int32_t r = tfhe_ctx_synchronize(k); if (r && !rc) rc = r; }
        return rc;
    }
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (hipSetDevice(c->device) != hipSuccess) return TFHE_ERR_DEVICE;
    if (c->stream && hipStreamSynchronize(c->stream) != hipSuccess) rc = TFHE_ERR_DEVICE;
    const hipStream_t ts = c->twin_stream.load(std::memory_order_acquire);
    if (ts && hipStreamSynchronize(ts) != hipSuccess) rc = TFHE_ERR_DEVICE;
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    (void)hipGetLastError();
    return rc;
}

int32_t tfhe_ctx_params(const tfhe_ctx *ctx, tfhe_params *out)
{
    if (!ctx || !out) return TFHE_ERR_INVALID_ARG;
    *out = ctx->P;
    return TFHE_OK;
}

}  // extern "C"
