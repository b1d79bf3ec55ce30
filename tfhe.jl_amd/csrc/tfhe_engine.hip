// tfhe_engine.hip — MI355X (gfx950) TFHE gate-bootstrapping engine: context, key loading, launchers, C ABI
// (include/tfhe_mi355x.h).  The kernels live in the headers included below; this file is the only
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
//
// A context is either a device context (one GPU: keys, workspaces, one stream) or a fan-out context created by
// tfhe_ctx_create_multi: it owns one device context per entry of device_ids[], replicates keys to all of them at
// load time and splits every host-buffer batch call into contiguous, rotation-balanced shards run concurrently on
// library-owned threads.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include "../../include/tfhe_mi355x.h"
#include "br_core.hpp"

using namespace tfhe;

#include "kernels_gates.hpp"
#include "kernels_blind_rotate.hpp"
#ifndef TFHE_NO_G2
#include "mk_g2_launch.hpp"
#endif
#include "kernels_anyn.hpp"
#include "kernels_n512.hpp"
#include "kernels_keyswitch.hpp"
#include "kernels_keygen.hpp"

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

// Rows of the wire table travelling from one device of a multi-device context to another (pull_wires).  A small ring of
// slots per ordered pair, each with its own buffers and events, so that a level can queue its transfers while those of the
// previous levels are still in flight; the host waits only if the ring wraps onto a slot whose transfer has not finished.
struct PairXfer {
    static constexpr int kSlots = 4;
    struct Slot {
        DevBuf out, idx_src;          // on the source device: the gathered rows, their wire indices
        DevBuf in, idx_dst;           // on the destination device: the rows as they arrive, their wire indices
        int32_t *h_idx = nullptr; size_t h_idx_cap = 0;     // pinned staging of the indices (read by both uploads)
        void *h_rows = nullptr; size_t h_rows_cap = 0;      // pinned staging of the rows (host path only)
        hipEvent_t ready = nullptr;   // recorded on the source's stream: rows gathered (host path: and copied out)
        hipEvent_t taken = nullptr;   // recorded on the destination's stream: rows scattered into its table
        bool pending = false;         // `taken` recorded and not yet known to have completed
    } slot[kSlots];
    unsigned next = 0;
};

struct tfhe_ctx {
    tfhe_params P{};
    int device = 0;
    std::string err;
    std::vector<tfhe_ctx *> kids;        // non-empty: fan-out context (tfhe_ctx_create_multi); no device state of its own
    std::vector<uint8_t> kid_ran;        // which kids took part in the last batch call (timing / diagnostics)

    hipStream_t stream = nullptr;
    // Timing events of the last kTimingSlots batch calls (batch start, BR start / end (= KS start), KS end): a caller that
    // times a sequence of calls reads them all afterwards (tfhe_timing_history_ms) instead of synchronising after each
    static constexpr int kTimingSlots = 33;      // 32 reportable sets + the one being recorded
    hipEvent_t evring[kTimingSlots][4] = {};
    hipEvent_t *ev = evring[0];          // the current call's set
    int64_t timed_calls = 0;             // batch calls that recorded a set so far
    hipEvent_t done_ev = nullptr;        // recorded at the end of every batch call: the workspaces are shared, so the next
    bool done_pending = false;           // call makes ITS stream wait for this event (no foreign stream handle is kept)
    bool timing_valid = false;
    bool own_pending = false;    // work queued on the context's own stream since the last done_ev (leave_stream)
    int64_t last_rotations = 0;
    std::string last_kernel;             // blind-rotate kernel instantiation the last batch call launched
    int ks_slices_large = 2;     // K-split of the MFMA keyswitch for large batches (tfhe_set_option("ks_slices", 1|2|4))
    int ks_variant = 4;          // 1 = one workgroup per sample, 3 = tiled + sliced + XCD-aware integer VALU, 4 = int8 MFMA (default)
    int ks_mode = 0;             // kernel family the loaded keyswitch key was laid out for (decided at load: pick_ks_mode)
    int64_t br_small = 1024;     // batches of at most this many rotations use the two-waves-per-rotation kernel (-1: never): what the chip holds at two waves per SIMD, 4 per CU (set at creation: 1024 on 256 CUs)
    int br_prio_pct = 90;        // a wave of the batched kernels lowers its issue priority 3 -> 0 over this share of its steps (0: off)
    int br_general = 0;          // tfhe_set_option("br_general", 1): every single-key blind rotation on blind_rotate_kernel_general (cross-check of the specialised kernels)
    int br_split = 1;            // tfhe_set_option("br_split", 0 | 1): batches above what the chip holds send their last, partly filled round (<= br_small rotations) to the two-waves-per-rotation kernels in a second launch (launch_blind_rotate)
    int timing_events = 1;       // 0: the gate entry points record no timing events (tfhe_last_timing_ms then has nothing to report)
    int br_rt_l = 0;             // 1: the run-time-l instantiations (L = 0) even for l = 2, 3 (A/B)
    int64_t br_tiny = -2;        // batches of at most this many rotations split every transform over two waves (-1: never; -2: one per CU =
                                 //  the device's CU count: 1.75 vs 1.93 ms up to 256 rotations at the 80-bit set, 2.6 vs 3.1 ms at the 128-bit set;
                                 //  3.5 vs 2.6 ms at 320 — profiles/r03/r03h2_*);
                                 // measured (interleaved A/B): 1 gate 1.83 vs 1.91 ms (l = 2), 2.76 vs 3.07 ms (l = 3); 32 gates: 2 % slower

    // tables
    cplx *d_tables = nullptr;   // tw1[512] | tw2[64] | twist[512]
    Tables T{};
    Gadget g{};
    // any-N kernels (kernels_anyn.hpp): e^{-2 pi i t/M} [M] | e^{-i pi j/N} [M] for THIS context's N
    cplx *d_anyn_tab = nullptr;
    int br_anyn = 0;             // tfhe_set_option("br_anyn", 1): the any-N kernel (and its key layout) even where a tuned kernel exists; before the key is loaded
    int anyn_spec = -1;          // any-N kernel: spectrum accumulators in LDS (0) / in global memory (1) / LDS when they fit (-1)
    // Parameter sets outside what the tuned kernels and blind_rotate_kernel_general are built for (N other than 1024 / 2048,
    // k > 4; multi-key: N other than 1024, more than 8 parties, l > 8) run on the any-N kernels, which need the key in their
    // own spectrum order: decided once, consulted by the loaders and the dispatcher
    bool anyn() const
    {
        if (br_anyn) return true;
        if (P.parties > 1) return P.N != kN || P.parties > 8 || P.bs_l > 8;
        if (n512()) return false;
        return (P.N != kN && P.N != 2048) || P.k > 4;
    }
    // N = 512 with k = 1 (any l) has a tuned kernel of its own (kernels_n512.hpp) and its own key order
    bool n512() const { return !br_anyn && P.parties == 1 && P.N == 512 && P.k == 1; }

    // keys (only the layout of the selected keyswitch kernel family stays resident)
    cplx *d_bk = nullptr;       size_t bk_polys = 0;
    int32_t *d_ks = nullptr;    // canonical [kN][t][base-1][n+1]                   (ks_mode 1)
    int32_t *d_ksp = nullptr;   int ks_stride = 0;   // row-padded copy             (ks_mode 3)
    void *d_ks4 = nullptr;      int ks4_wtiles = 0;  // MFMA B fragments            (ks_mode 4: base 4, t = 8)
    void *d_mk_ks4 = nullptr;   size_t mk_ks4_frags = 0;
    bool have_bk = false, have_ks = false;
    // multi-key
    cplx *d_mk_bk = nullptr;
    int32_t *d_mk_ksp = nullptr;   // [P] row-padded keyswitch keys back to back
    size_t mk_ksp_words = 0;       // words per party in d_mk_ksp
    int mk_parties = 0;
    bool have_mk_bk = false, have_mk_ks = false;

    // device-resident wire table for levelised circuits: int32 [num_wires][n+1]
    int32_t *d_wires = nullptr; int64_t num_wires = 0;

    // workspaces
    DevBuf bara, ext, map, io[4], diag, abar, mk_acc, spec;
    size_t diag_rows = 0;
    bool mk_force_general = false; // tfhe_set_option("mk_general", 1): use the any-P kernel for 2 parties too (cross-check)
    int n2048_rw = 0;              // N = 2048: rotations per workgroup advancing in lockstep (tfhe_set_option("n2048_rw", 0|1|2); 0 = one up to
                                   //  one rotation per CU — the pair would leave half the CUs idle: 7.9 vs 9.0 ms at 64 rotations — two beyond)
    int mkg_rw = 0;                // any-party kernel: rotations per workgroup, in lockstep (0: two; otherwise a cap, at most 4 and what fits in LDS)
    int mkg_variant = 0;           // 4- / 8-party shipped sets: 0 = two-wave kernel with compile-time (parties, l), 1 = the any-party kernel
    int mkg_acc = -1;              // any-party kernel: accumulators in LDS (0) / in global memory (1) / by party count (-1: global above 4 parties)
    int mk_rw = 0;                 // two-wave 2-party kernel: rotations per workgroup advancing in lockstep (tfhe_set_option("mk_rw", 0|1|2); 0 = one
                                   //  up to one rotation per CU (single mk_gate_nand 11.7 vs 13.7 ms), two beyond)
    bool measure_margin = false;   // tfhe_set_option("measure_margin", 1): blind rotations run their DIAG instantiation
    // Host-buffer batches of at least `pipeline_min` gates are cut in two rotation-balanced halves that run on two streams of
    // this device, so that the second half's upload and the first half's download cross PCIe while the other half computes.
    // The second stream, its workspaces and its events belong to a twin context that BORROWS this context's keys.
    tfhe_ctx *twin = nullptr;
    bool borrows_keys = false;
    bool slot_busy[2] = {false, false};   // tfhe_gates_batch_submit: a batch is in flight on the own (0) / the twin's (1) stream
    uint32_t submits = 0;
    // multi-device context: coherence of the replicated wire table.  wire_valid[k][w]: device k's replica holds wire w's
    // current value; wire_owner[w]: a device that does (the one that wrote it last).  A level's outputs become valid on the
    // device that computed them only; whoever reads them elsewhere later fetches them then (pull_wires), device to device.
    std::vector<std::vector<uint8_t>> wire_valid;
    std::vector<int32_t> wire_owner;
    std::vector<uint8_t> peer_ok;         // [nk * nk]: device-to-device copies allowed between kids' devices (hipDeviceCanAccessPeer, or the same device)
    std::vector<struct PairXfer *> xfer;  // [nk * nk] rows in flight from kid src to kid dst (made on first use)
    int level_exchange = 0;               // multi-device context: how a sharded level's rows reach the other replicas: 0 = device-to-device copies where hipDeviceCanAccessPeer allows (else pinned host staging), 1 = device-to-device, 2 = host staging
    int64_t level_split_min = 4096;       // multi-device context: levels of at least this many blind rotations are sharded over the devices (tfhe_set_option("level_split_min", n); < 0: never)
    std::vector<int32_t> kid_tickets[2];  // multi-device context: per submit slot, the ticket every kid gave for its shard (2: none)
    int cu_count = 256;          // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    int w2_rw = 0;               // tfhe_set_option("w2_rw", 0 | 1 | 2): rotations per workgroup of the two-wave kernel; 0 = pairs up to two rotations per CU and at (nearly) four
    int n512_w2 = -1;            // tfhe_set_option("n512_w2", -1 | 0 | 1): the two-waves-per-rotation N = 512 kernel up to 6 rotations per CU (-1), never, always
    int n512_rw = 0;             // tfhe_set_option("n512_rw", 0 | 1 | 4): rotations per workgroup of the N = 512 kernel (0: by batch size)
    int k2_w3 = -1;              // tfhe_set_option("k2_w3", -1 | 0 | 1): the three-waves-per-rotation k = 2 kernel for batches of up to two rotations per CU and for the last round of a larger one (-1: by size), never (0), for every batch (1)
    int k2_rw = 0;               // tfhe_set_option("k2_rw", 0 | 1 | 7): rotations per workgroup of the k = 2 kernel; 0 = equally full rounds of up to seven per CU
    int v3_rw = 0;               // tfhe_set_option("v3_rw", 0 | 1 | 4): rotations per workgroup of the default kernel; 0 = 4 from 1536 rotations up
    int64_t pipeline_min = 4096;   // tfhe_set_option("pipeline_min", n); < 0: never
    bool last_call_two_streams = false;   // the last batch call ran as two halves: timings span both streams
    // Pinned staging for the index maps of a call, a ring of kMapStages blocks: the H2D copy of a call's maps sits in the stream
    // behind the previous call's kernels, so with ONE block the host would wait for the previous call to finish before it could
    // fill in the next one's (a circuit level per call: the host never ran ahead of the device); with four it queues up to three
    // calls ahead.  ensure_host_map() hands out the next block (h_map / map_ev / map_cur point at it).
    struct MapStage { void *h = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
    static constexpr int kMapStages = 4;
    MapStage map_stage[kMapStages];
    unsigned map_next = 0;
    MapStage *map_cur = &map_stage[0];
    void *h_map = nullptr;                 // = map_cur->h
    hipEvent_t map_ev = nullptr;           // = map_cur->ev: recorded behind the H2D copy of the block

    // "calls on one context must not overlap": the thread inside an entry point owns the context (CallGuard below); a second
    // thread's overlapping call gets TFHE_ERR_STATE instead of racing on the shared workspaces
    std::atomic<std::thread::id> owner{};
    int owner_depth = 0;

    bool multi() const { return !kids.empty(); }

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


// ---- one caller at a time ------------------------------------------------------------------------------------------
// Every entry point that takes a (non-const) context enters through CallGuard.  The first thread in becomes the owner until its
// outermost call returns (entry points call one another: the owner may nest); any other thread's call fails with TFHE_ERR_STATE
// and a message that tfhe_last_error() returns to THAT thread (the context's own message string belongs to the owner).
// Streaming stays what it was: tfhe_gates_batch_submit returns while the batch runs on the device; the guard covers the host side.
static thread_local const tfhe_ctx *g_rejected_ctx = nullptr;
static const char kBusyMessage[] = "the context is inside another call on another thread: calls on one context must not overlap (one caller at a time; use one context per thread)";
struct CallGuard {
    tfhe_ctx *c;
    bool ok;
    explicit CallGuard(tfhe_ctx *c_) : c(c_), ok(false)
    {
        const std::thread::id me = std::this_thread::get_id();
        if (c->owner.load(std::memory_order_acquire) == me) { c->owner_depth++; ok = true; return; }
        std::thread::id none{};
        ok = c->owner.compare_exchange_strong(none, me, std::memory_order_acq_rel);
        if (ok) { c->owner_depth = 1; if (g_rejected_ctx == c) g_rejected_ctx = nullptr; }
        else g_rejected_ctx = c;
    }
    ~CallGuard()
    {
        if (ok && --c->owner_depth == 0) c->owner.store(std::thread::id{}, std::memory_order_release);
    }
    CallGuard(const CallGuard &) = delete;
    CallGuard &operator=(const CallGuard &) = delete;
};
#define ENTER_CTX(ctx)                                                                             \
    if (!(ctx)) return TFHE_ERR_INVALID_ARG;                                                       \
    CallGuard call_guard_(ctx);                                                                    \
    if (!call_guard_.ok) return TFHE_ERR_STATE

// ---- hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel), never lowered ---------------------------
// A driver call on the host side of every launch of a kernel that needs more than 64 KB of LDS — on the path where latency is
// the metric (a single gate, a circuit level).  The attribute belongs to (device, function), is shared by every context of the
// process and only ever needs to grow (the any-N kernels' LDS depends on the parameter set), so the largest value set so far
// is remembered process-wide.
static int32_t ensure_dyn_lds(tfhe_ctx *c, const void *fn, size_t bytes, const char *what)
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
#define LDS_TRY(ctx, bytes, ...)                                                                   \
    do {                                                                                           \
        const int32_t rl_ = ensure_dyn_lds(ctx, (const void *)(__VA_ARGS__), (size_t)(bytes), #__VA_ARGS__); \
        if (rl_) return rl_;                                                                       \
    } while (0)

constexpr size_t kH2TableOffset = kTableElems + 1024;      // tw1h | tw2q | tw3q of blind_rotate_kernel_h2
constexpr size_t kN512TableOffset = kH2TableOffset + kH2TableElems;      // tw1 of blind_rotate_kernel_n512

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

static int ilog2i(int x) { int r = 0; while ((1 << r) < x) r++; return r; }

// ---- sharding of a gate stream: contiguous shards balanced by blind-rotation count ----------
// MUX = 2 rotations, NOT / CONST / COPY = 0 (gates.jl:163-177, 76-93); every gate also weighs 1/1000 so that cut
// points stay well defined among trivial gates.  bounds[r] .. bounds[r+1] is shard r.  Same rule as
// tfhe.jl_amd/sharding.py:shard_bounds (tests compare the two).
static void shard_bounds_by_rotations(const uint8_t *opcodes, int64_t B, int shards, int64_t *bounds)
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

int32_t tfhe_abi_version(void) { return TFHE_MI355X_ABI_VERSION; }

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
    return ctx->err.c_str();
}

int32_t tfhe_shard_bounds(const uint8_t *opcodes, int64_t B, int32_t shards, int64_t *bounds)
{
    if (B < 0 || shards < 1 || !bounds) return TFHE_ERR_INVALID_ARG;
    shard_bounds_by_rotations(opcodes, B, shards, bounds);
    return TFHE_OK;
}

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
    *out_ctx = c;
    return TFHE_OK;
}

int32_t tfhe_ctx_create_multi(const tfhe_params *params, const int32_t *device_ids, int32_t n_dev, tfhe_ctx **out_ctx)
{
    if (!params || !out_ctx || !device_ids) { g_create_error = "tfhe_ctx_create_multi: NULL argument"; return TFHE_ERR_INVALID_ARG; }
    *out_ctx = nullptr;
    if (n_dev < 1 || n_dev > 64) { g_create_error = "tfhe_ctx_create_multi: n_dev must be 1..64"; return TFHE_ERR_INVALID_ARG; }
    tfhe_ctx *c = new tfhe_ctx();
    c->P = *params;
    c->device = device_ids[0];
    for (int i = 0; i < n_dev; i++) {
        tfhe_ctx *k = nullptr;
        const int32_t rc = tfhe_ctx_create(params, device_ids[i], &k);
        if (rc) {               // g_create_error holds the reason
            tfhe_ctx_destroy(c);
            return rc;
        }
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
    *out_ctx = c;
    return TFHE_OK;
}

int32_t tfhe_ctx_device_count(const tfhe_ctx *ctx) { return !ctx ? -1 : ctx->multi() ? (int32_t)ctx->kids.size() : 1; }

void tfhe_ctx_destroy(tfhe_ctx *c)
{
    if (!c) return;
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

int32_t tfhe_host_alloc(size_t bytes, void **out_ptr)
{
    if (!out_ptr) return TFHE_ERR_INVALID_ARG;
    *out_ptr = nullptr;
    if (bytes == 0) return TFHE_OK;
    const hipError_t e = hipHostMalloc(out_ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { g_create_error = std::string("tfhe_host_alloc: ") + hipGetErrorString(e); *out_ptr = nullptr; return TFHE_ERR_DEVICE; }
    return TFHE_OK;
}

void tfhe_host_free(void *ptr) { if (ptr) (void)hipHostFree(ptr); }

int32_t tfhe_ctx_params(const tfhe_ctx *ctx, tfhe_params *out)
{
    if (!ctx || !out) return TFHE_ERR_INVALID_ARG;
    *out = ctx->P;
    return TFHE_OK;
}

}  // extern "C"

__global__ void gather_rows_kernel(const int32_t *__restrict__ table, const int32_t *__restrict__ idx, int32_t *__restrict__ out, int n1)
{
    const size_t src = (size_t)idx[blockIdx.x], dst = blockIdx.x;
    for (int i = threadIdx.x; i < n1; i += blockDim.x) out[dst * n1 + i] = table[src * n1 + i];
}

__global__ void scatter_rows_kernel(const int32_t *__restrict__ rows, const int32_t *__restrict__ idx, int32_t *__restrict__ table, int n1)
{
    const size_t src = blockIdx.x, dst = (size_t)idx[blockIdx.x];
    for (int i = threadIdx.x; i < n1; i += blockDim.x) table[dst * n1 + i] = rows[src * n1 + i];
}

// ---- fan-out helpers --------------------------------------------------------------------------------
// Runs fn(kid index) for every kid in `which` concurrently (one library-owned thread per extra kid; the calling
// thread takes the first) and returns the first failing status, copying that kid's message.
template <typename F>
static int32_t fan_out(tfhe_ctx *c, const std::vector<int> &which, F &&fn)
{
    std::vector<int32_t> rcs(which.size(), TFHE_OK);
    std::vector<std::thread> th;
    std::vector<size_t> inline_ones;          // kids whose thread could not be started (thread / process limit): run here, in turn
    for (size_t i = 1; i < which.size(); i++) {
        try {
            th.emplace_back([&, i] { rcs[i] = fn(which[i]); });
        } catch (const std::exception &) {    // no exception may cross the C ABI
            inline_ones.push_back(i);
        }
    }
    if (!which.empty()) rcs[0] = fn(which[0]);
    for (size_t i : inline_ones) rcs[i] = fn(which[i]);
    for (auto &t : th) t.join();
    std::fill(c->kid_ran.begin(), c->kid_ran.end(), 0);
    for (int k : which) c->kid_ran[(size_t)k] = 1;
    for (size_t i = 0; i < which.size(); i++)
        if (rcs[i]) return c->set_err(rcs[i], "device %d (kid %d): %s", c->kids[(size_t)which[i]]->device, which[i], c->kids[(size_t)which[i]]->err.c_str());
    return TFHE_OK;
}
static std::vector<int> all_kids(const tfhe_ctx *c)
{
    std::vector<int> v(c->kids.size());
    for (size_t i = 0; i < v.size(); i++) v[i] = (int)i;
    return v;
}

// equal contiguous split of B rows for the entry points whose rows all cost the same
template <typename F>
static int32_t multi_rows(tfhe_ctx *c, int64_t B, F &&call)
{
    const int nk = (int)c->kids.size();
    std::vector<int> which;
    std::vector<int64_t> lo((size_t)nk), hi((size_t)nk);
    for (int r = 0; r < nk; r++) {
        lo[(size_t)r] = B * r / nk; hi[(size_t)r] = B * (r + 1) / nk;
        if (hi[(size_t)r] > lo[(size_t)r]) which.push_back(r);
    }
    return fan_out(c, which, [&](int r) { return call(c->kids[(size_t)r], lo[(size_t)r], hi[(size_t)r] - lo[(size_t)r]); });
}


extern "C" {

static size_t bk_poly_count(const tfhe_params &p) { return (size_t)p.n * p.bs_l * (p.k + 1) * (p.k + 1); }

// A key source may be a host buffer or (tfhe_keygen_cloud_key) a buffer on THIS context's device.  A buffer on another GPU
// is refused: copying from it would depend on peer access between the two devices (a multi-device context enables it only where
// hipDeviceCanAccessPeer allows, for its wire-table exchange).
static int32_t check_key_source(tfhe_ctx *c, const void *p, const char *who)
{
    hipPointerAttribute_t at;
    const hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return TFHE_OK; }       // an ordinary (unregistered) host pointer
    if (at.type == hipMemoryTypeDevice && at.device != c->device)
        return c->set_err(TFHE_ERR_DEVICE, "%s: the key buffer lives on device %d, this context on device %d: pass a host copy (peer access between GPUs is not assumed)",
                          who, at.device, c->device);
    return TFHE_OK;
}

// Before a key buffer is freed or replaced: nothing this context (or its second-stream twin) enqueued may still be running.
static void quiesce(tfhe_ctx *c)
{
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->done_pending) { (void)hipEventSynchronize(c->done_ev); c->done_pending = false; }      // a call still running on a caller's stream
    c->own_pending = false;
    if (c->twin && c->twin->stream) (void)hipStreamSynchronize(c->twin->stream);
    c->slot_busy[0] = c->slot_busy[1] = false;
    if (c->twin) {
        // every caller of quiesce is about to free or replace key buffers: the twin borrows them, so it must not keep the old
        // addresses (ensure_twin re-points it at the owner's current keys before its next use)
        tfhe_ctx *t = c->twin;
        t->d_bk = nullptr; t->d_ks = nullptr; t->d_ksp = nullptr; t->d_ks4 = nullptr; t->have_bk = false; t->have_ks = false;
    }
}

static int32_t load_bk_common(tfhe_ctx *c, const void *host, size_t bytes_in, bool is_c128)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!host) return c->set_err(TFHE_ERR_INVALID_ARG, "load_bootstrap_key: NULL key pointer");
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "load_bootstrap_key: context is multi-key, use tfhe_mk_load_*");
    if (c->multi()) return fan_out(c, all_kids(c), [&](int k) { return load_bk_common(c->kids[(size_t)k], host, bytes_in, is_c128); });
    HIP_TRY(c, hipSetDevice(c->device));
    { const int32_t rcp = check_key_source(c, host, "load_bootstrap_key"); if (rcp) return rcp; }
    const size_t npolys = bk_poly_count(c->P);
    const bool big = (c->P.N == kN2);
    quiesce(c);
    if (c->d_bk) { (void)hipFree(c->d_bk); c->d_bk = nullptr; c->have_bk = false; }
    HIP_TRY(c, hipMalloc((void **)&c->d_bk, npolys * (size_t)(c->P.N / 2) * sizeof(cplx)));
    void *d_in = nullptr;
    HIP_TRY(c, hipMalloc(&d_in, bytes_in));
    hipError_t e = hipMemcpyAsync(d_in, host, bytes_in, hipMemcpyDefault, c->stream)   /* host pointer, or a device buffer (tfhe_keygen_cloud_key) */;
    if (e == hipSuccess && c->n512()) {
        const cplx *t1 = c->d_tables + kN512TableOffset, *t2 = c->d_tables + kH2TableOffset + 512, *t3 = t2 + 64;
        if (is_c128) hipLaunchKernelGGL(bk_permute_c128_kernel_n512, dim3((unsigned)npolys), dim3(64), 0, c->stream, (const cplx *)d_in, c->d_bk);
        else hipLaunchKernelGGL(bk_prepare_kernel_n512, dim3((unsigned)npolys), dim3(64), 0, c->stream, (const int32_t *)d_in, c->d_bk, t1, t2, t3);
        e = hipGetLastError();
    } else if (e == hipSuccess && c->anyn()) {
        // the any-N kernels' spectrum order (kernels_anyn.hpp): the same forward transform they run, or a permutation of the reference's spectra
        const int log2N = ilog2i(c->P.N), M = c->P.N / 2;
        if (is_c128) {
            const size_t total = npolys * (size_t)M;
            hipLaunchKernelGGL(anyn::bk_permute_c128_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, (const cplx *)d_in, c->d_bk, log2N - 1, total);
        } else {
            const size_t ldsp = (size_t)anyn::padded_len(M > 0 ? M : 1) * sizeof(cplx);
            if (ldsp > 64 * 1024 && ensure_dyn_lds(c, (const void *)anyn::bk_prepare_kernel, ldsp, "anyn::bk_prepare_kernel")) e = hipErrorInvalidValue;
            if (e == hipSuccess)
                hipLaunchKernelGGL(anyn::bk_prepare_kernel, dim3((unsigned)npolys), dim3((unsigned)anyn::threads_for(c->P.N)), ldsp, c->stream, (const int32_t *)d_in, c->d_bk,
                                   (const cplx *)c->d_anyn_tab, (const cplx *)(c->d_anyn_tab + M), log2N, 1.0 / (double)M);
        }
        if (e == hipSuccess) e = hipGetLastError();
    } else if (e == hipSuccess) {
        if (big && is_c128)
            hipLaunchKernelGGL(bk_permute_c128_kernel_n2048, dim3((unsigned)npolys), dim3(128), 0, c->stream, (const cplx *)d_in, c->d_bk);
        else if (big)
            hipLaunchKernelGGL(bk_prepare_kernel_n2048, dim3((unsigned)npolys), dim3(128), 0, c->stream, (const int32_t *)d_in, c->d_bk,
                               (const cplx *)(c->d_tables + kTableElems), c->T.tw2);
        else if (is_c128)
            hipLaunchKernelGGL(bk_permute_c128_kernel, dim3((unsigned)npolys), dim3(64), 0, c->stream, (const cplx *)d_in, c->d_bk);
        else
            hipLaunchKernelGGL(bk_prepare_kernel, dim3((unsigned)npolys), dim3(64), 0, c->stream, (const int32_t *)d_in, c->d_bk, c->T, 1.0 / kM);
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
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    return load_bk_common(c, bk, bk_poly_count(c->P) * (size_t)c->P.N * sizeof(int32_t), false);
}

int32_t tfhe_load_bootstrap_key_c128(tfhe_ctx *c, const double *bk_spectra)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    return load_bk_common(c, bk_spectra, bk_poly_count(c->P) * (size_t)(c->P.N / 2) * sizeof(cplx), true);
}

static size_t ks_word_count(const tfhe_params &p)
{
    return (size_t)p.k * p.N * p.ks_t * ((1u << p.ks_log2_base) - 1) * (size_t)(p.n + 1);
}

// Which keyswitch kernel family serves this context (decided when the key is loaded, so that only that family's key
// layout stays resident): 4 = int8 MFMA (base 4, t = 8), 3 = tiled integer VALU (base 4, t multiple of 4), 1 = gather.
static int pick_ks_mode(const tfhe_ctx *c)
{
    const int kNn = c->P.k * c->P.N;
    const bool ok4 = c->P.ks_log2_base == 2 && c->P.ks_t == 8 && kNn % 128 == 0;
    const bool ok3 = c->P.ks_log2_base == 2 && c->P.ks_t % 4 == 0 && kNn % KS3_SLICES == 0 && kNn / KS3_SLICES <= 128;
    if (c->ks_variant == 4 && ok4) return 4;
    if (c->ks_variant >= 3 && ok3) return 3;
    return 1;
}

int32_t tfhe_load_keyswitch_key(tfhe_ctx *c, const int32_t *ks)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!ks) return c->set_err(TFHE_ERR_INVALID_ARG, "load_keyswitch_key: NULL key pointer");
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "load_keyswitch_key: context is multi-key, use tfhe_mk_load_*");
    if (c->multi()) return fan_out(c, all_kids(c), [&](int k) { return tfhe_load_keyswitch_key(c->kids[(size_t)k], ks); });
    HIP_TRY(c, hipSetDevice(c->device));
    { const int32_t rcp = check_key_source(c, ks, "load_keyswitch_key"); if (rcp) return rcp; }
    const size_t bytes = ks_word_count(c->P) * sizeof(int32_t);
    c->have_ks = false;
    quiesce(c);
    if (c->d_ks) { (void)hipFree(c->d_ks); c->d_ks = nullptr; }
    if (c->d_ksp) { (void)hipFree(c->d_ksp); c->d_ksp = nullptr; }
    if (c->d_ks4) { (void)hipFree(c->d_ks4); c->d_ks4 = nullptr; }
    int32_t *d_canon = nullptr;
    HIP_TRY(c, hipMalloc((void **)&d_canon, bytes));
    const int mode = pick_ks_mode(c);
    auto body = [&]() -> int32_t {
        // host pointer, or a device buffer (tfhe_keygen_cloud_key).  On the context's stream, like everything that consumes
        // d_canon below: a device-to-device hipMemcpy is NOT synchronous with the host and runs on the NULL stream, which
        // this context's non-blocking stream does not wait for
        HIP_TRY(c, hipMemcpyAsync(d_canon, ks, bytes, hipMemcpyDefault, c->stream));
        if (mode == 3) {   // row-padded copy: stride = n+1 rounded up to 4 words so that rows are 16-byte aligned
            const size_t n1 = (size_t)c->P.n + 1, stride = (n1 + 3) & ~(size_t)3;
            const size_t rows = ks_word_count(c->P) / n1;
            HIP_TRY(c, hipMalloc((void **)&c->d_ksp, rows * stride * 4));
            HIP_TRY(c, hipMemsetAsync(c->d_ksp, 0, rows * stride * 4, c->stream));
            HIP_TRY(c, hipMemcpy2DAsync(c->d_ksp, stride * 4, d_canon, n1 * 4, n1 * 4, rows, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->ks_stride = (int)stride;
        } else if (mode == 4) {
            const int kNn = c->P.k * c->P.N, wtiles = (c->P.n + 1 + 31) / 32;
            const size_t frags = (size_t)kNn * wtiles * 4 * 64;
            HIP_TRY(c, hipMalloc(&c->d_ks4, frags * 16));
            hipLaunchKernelGGL(ks4_prepare_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, c->stream, (const int32_t *)d_canon,
                               (i32x4 *)c->d_ks4, c->P.n, kNn, wtiles);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            c->ks4_wtiles = wtiles;
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));      // the caller's buffer is free again, the key is complete for any stream
        return TFHE_OK;
    };
    const int32_t rc = body();
    if (rc == TFHE_OK && mode == 1) c->d_ks = d_canon;       // the gather kernel reads the canonical layout
    else (void)hipFree(d_canon);
    if (rc) return rc;
    c->ks_mode = mode;
    c->have_ks = true;
    return TFHE_OK;
}

// Generates the cloud key on the device (kernels_keygen.hpp) and loads it: the analogue of CloudKey(rng, secret_key)
// (api.jl:111-127) with the secret material supplied by the caller.  Optionally copies the canonical Int32 arrays back.
int32_t tfhe_keygen_cloud_key(tfhe_ctx *c, const int32_t *lwe_key, const int32_t *tlwe_key, double bs_noise_stddev,
                              double ks_noise_stddev, const uint32_t *seed, int32_t *bk_out, int32_t *ks_out)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!lwe_key || !tlwe_key || !seed) return c->set_err(TFHE_ERR_INVALID_ARG, "keygen_cloud_key: NULL key or seed pointer");
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "keygen_cloud_key: context is multi-key (use tfhe_mk_expand_load_bootstrap_key)");
    if (!(bs_noise_stddev >= 0.0) || !(ks_noise_stddev >= 0.0)) return c->set_err(TFHE_ERR_INVALID_ARG, "keygen_cloud_key: negative noise parameter");
    tfhe_ctx *g = c->multi() ? c->kids[0] : c;             // a fan-out context generates on its first device
    HIP_TRY(c, hipSetDevice(g->device));
    const tfhe_params &P = c->P;
    const size_t bk_words = bk_poly_count(P) * (size_t)P.N, ks_words = ks_word_count(P);
    const size_t Q = ks_words / (size_t)(P.n + 1), kN = (size_t)P.k * P.N;
    int32_t *d_lwe = nullptr, *d_tlwe = nullptr, *d_bk = nullptr, *d_ks = nullptr;
    double *d_noise = nullptr;
    // secret material (the key bits, the raw noise) is zeroed before its memory goes back to the allocator, on the success and
    // on every error path
    auto cleanup = [&]() {
        if (d_lwe) (void)hipMemsetAsync(d_lwe, 0, (size_t)P.n * 4, g->stream);
        if (d_tlwe) (void)hipMemsetAsync(d_tlwe, 0, kN * 4, g->stream);
        if (d_noise) (void)hipMemsetAsync(d_noise, 0, (Q + 1) * sizeof(double), g->stream);
        (void)hipStreamSynchronize(g->stream);
        (void)hipFree(d_lwe); (void)hipFree(d_tlwe); (void)hipFree(d_bk); (void)hipFree(d_ks); (void)hipFree(d_noise);
    };
    auto body = [&]() -> int32_t {
        HIP_TRY(c, hipMalloc((void **)&d_lwe, (size_t)P.n * 4));
        HIP_TRY(c, hipMalloc((void **)&d_tlwe, kN * 4));
        HIP_TRY(c, hipMalloc((void **)&d_bk, bk_words * 4));
        HIP_TRY(c, hipMalloc((void **)&d_ks, ks_words * 4));
        HIP_TRY(c, hipMalloc((void **)&d_noise, (Q + 1) * sizeof(double)));
        HIP_TRY(c, hipMemcpyAsync(d_lwe, lwe_key, (size_t)P.n * 4, hipMemcpyHostToDevice, g->stream));
        HIP_TRY(c, hipMemcpyAsync(d_tlwe, tlwe_key, kN * 4, hipMemcpyHostToDevice, g->stream));
        keygen::Args A;
        A.lwe_key = d_lwe; A.tlwe_key = d_tlwe; A.bk = d_bk; A.ks = d_ks; A.ks_noise = d_noise; A.ks_mean = d_noise + Q;
        A.n = P.n; A.N = P.N; A.k = P.k; A.l = P.bs_l; A.beta = P.bs_log2_base; A.t = P.ks_t; A.ks_log2_base = P.ks_log2_base;
        A.bs_alpha = bs_noise_stddev; A.ks_alpha = ks_noise_stddev;
        for (int i = 0; i < 6; i++) A.seed.w[i] = seed[i];
        const size_t samples = (size_t)P.n * P.bs_l * (P.k + 1);
        const size_t lds = kN * 4 + (size_t)P.k * ((P.N + 31) / 32) * 4;
        hipLaunchKernelGGL(keygen::bk_kernel, dim3((unsigned)samples), dim3(256), lds, g->stream, A);
        hipLaunchKernelGGL(keygen::ks_noise_kernel, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, g->stream, A, Q);
        hipLaunchKernelGGL(keygen::ks_mean_kernel, dim3(1), dim3(256), 0, g->stream, A, Q);
        hipLaunchKernelGGL(keygen::ks_kernel, dim3((unsigned)((Q + 3) / 4)), dim3(256), 0, g->stream, A, Q);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(g->stream));
        // A context with devices other than the generating one replicates through HOST copies (the caller's bk_out / ks_out,
        // or a staging vector): the loaders then see plain host pointers on every device and nothing depends on peer
        // access between GPUs being enabled.  One device: the loaders copy straight from the generated device buffers.
        bool other_device = false;
        if (c->multi())
            for (const tfhe_ctx *k : c->kids) other_device = other_device || k->device != g->device;
        std::vector<int32_t> stage_bk, stage_ks;
        if (other_device && !bk_out) { stage_bk.resize(bk_words); bk_out = stage_bk.data(); }
        if (other_device && !ks_out) { stage_ks.resize(ks_words); ks_out = stage_ks.data(); }
        if (bk_out) HIP_TRY(c, hipMemcpyAsync(bk_out, d_bk, bk_words * 4, hipMemcpyDeviceToHost, g->stream));
        if (ks_out) HIP_TRY(c, hipMemcpyAsync(ks_out, d_ks, ks_words * 4, hipMemcpyDeviceToHost, g->stream));
        HIP_TRY(c, hipStreamSynchronize(g->stream));
        int32_t rc = tfhe_load_bootstrap_key_i32(c, other_device ? bk_out : d_bk);
        if (rc) return rc;
        return tfhe_load_keyswitch_key(c, other_device ? ks_out : d_ks);
    };
    const int32_t rc = body();
    (void)hipSetDevice(g->device);
    cleanup();
    return rc;
}

// ---- launch helpers ------------------------------------------------------------------------------
static int32_t prepare_diag(tfhe_ctx *c, size_t R, hipStream_t s, DiagArgs &d)
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

static void name_kernel(tfhe_ctx *c, const char *fmt, ...)
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
        const size_t ldsb = (size_t)rw * (2 * kImg2 * 4 + 2 * kXchElems * sizeof(cplx)) + 64 * sizeof(cplx);
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
        const size_t ldsh = 2 * kImg * 4 + (size_t)4 * L * (kH2Buf + 256) * sizeof(cplx);
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
static double kK2W3RoundCost[3] = {0.0, 3.1, 4.3};
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

static int32_t launch_blind_rotate(tfhe_ctx *c, size_t R, int32_t mu, hipStream_t s)
{
    DiagArgs diag;
    int32_t rc = prepare_diag(c, R, s, diag);
    if (rc) return rc;
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

// Workspaces (bara, ext, abar, map, diag) are per context: work a previous call enqueued (possibly on another stream)
// must have finished before a call on stream `s` reuses them.  The previous call recorded done_ev at its end; making
// `s` wait for that event orders the two without blocking the host and without keeping the caller's stream handle.
static int32_t enter_stream(tfhe_ctx *c, hipStream_t s)
{
    if (c->own_pending && s != c->stream) {
        // the previous call ran on the context's own stream and recorded nothing (leave_stream): record now, for this caller's stream
        HIP_TRY(c, hipEventRecord(c->done_ev, c->stream));
        c->own_pending = false;
        c->done_pending = true;
    }
    if (!c->done_pending) return TFHE_OK;
    // Already finished (the common case for callers that synchronise between calls): nothing to order, and the event is not
    // handed to the runtime again — the stream it was recorded on may have been destroyed by its owner since.
    const hipError_t q = hipEventQuery(c->done_ev);
    if (q == hipSuccess) { c->done_pending = false; return TFHE_OK; }
    if (q != hipErrorNotReady) return c->set_err(TFHE_ERR_DEVICE, "hipEventQuery failed: %s", hipGetErrorString(q));
    HIP_TRY(c, hipStreamWaitEvent(s, c->done_ev, 0));
    return TFHE_OK;
}
static int32_t leave_stream(tfhe_ctx *c, hipStream_t s)
{
    // Calls that follow one another on the context's own stream are ordered by the stream; the event exists for a caller that
    // changes streams between calls, and is then recorded on demand (enter_stream).  (An event record keeps the next kernel of the
    // stream waiting ~5 us: six of them per circuit level were 1.3 % of the tutorial circuit.)
    if (s == c->stream) { c->own_pending = true; return TFHE_OK; }
    c->own_pending = false;
    HIP_TRY(c, hipEventRecord(c->done_ev, s));
    c->done_pending = true;
    return TFHE_OK;
}

// Every batch call records its four timing events into the next slot of the ring.  The slot is TAKEN at the start of a call
// and COUNTED only once its fourth event has been recorded (commit_timing_slot): a call that fails between the two leaves
// timed_calls alone, the next call reuses the slot, and tfhe_timing_history_ms never sees a half-recorded set.
static void next_timing_slot(tfhe_ctx *c)
{
    c->ev = c->evring[c->timed_calls % tfhe_ctx::kTimingSlots];
    c->timing_valid = false;
    c->last_call_two_streams = false;
}
static void commit_timing_slot(tfhe_ctx *c)
{
    c->timed_calls++;
    c->timing_valid = true;
}

static int32_t ensure_host_map(tfhe_ctx *c, size_t bytes)
{
    tfhe_ctx::MapStage &st = c->map_stage[c->map_next++ % tfhe_ctx::kMapStages];
    if (st.pending) {   // the H2D copy of the call that used this block (four calls ago) must have been consumed
        HIP_TRY(c, hipEventSynchronize(st.ev));
        st.pending = false;
    }
    if (bytes > st.cap) {
        if (st.h) (void)hipHostFree(st.h);
        st.h = nullptr; st.cap = 0;
        HIP_TRY(c, hipHostMalloc(&st.h, bytes + bytes / 4 + 256, hipHostMallocDefault));
        st.cap = bytes + bytes / 4 + 256;
    }
    c->map_cur = &st;
    c->h_map = st.h;
    c->map_ev = st.ev;
    return TFHE_OK;
}

static inline bool op_has_a(int op) { return !(op == TFHE_GATE_CONST0 || op == TFHE_GATE_CONST1); }
static inline bool op_has_b(int op) { return op_has_a(op) && !(op == TFHE_GATE_NOT || op == TFHE_GATE_COPY); }

// Common body of tfhe_gates_batch_dev (operands = rows g of three arrays, ia = ib = ic = io = NULL) and
// tfhe_gates_level (operands = rows ia[g], ib[g], ic[g] of one wire table, result row io[g]; operands an opcode
// does not read are replaced by row 0).
static int32_t run_gates(tfhe_ctx *c, const char *who, const uint8_t *opcodes, int64_t B, const int32_t *d_in0,
                         const int32_t *d_in1, const int32_t *d_in2, int32_t *d_out, const int32_t *ia, const int32_t *ib,
                         const int32_t *ic, const int32_t *io, hipStream_t s)
{
    // development aid (TFHE_DEBUG_HOSTTIME=1): host microseconds per section of this function, printed per call — what showed that a
    // circuit level's host time was the wait for the previous level's staging copy, not anything in here (round 5)
    static const bool dbg_host = getenv("TFHE_DEBUG_HOSTTIME") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    double t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto MARK = [&](int k) { if (dbg_host) { const auto t = std::chrono::steady_clock::now(); t_acc[k] += std::chrono::duration<double, std::micro>(t - t_prev).count(); t_prev = t; } };
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
    {
        const int32_t rc0 = enter_stream(c, s);
        if (rc0) return rc0;
    }

    // index maps, one pinned staging block:
    //   rot_a[R] | rot_b[R] | ks_e0[G] | ks_e1[G] | ks_dst[G] | triv_src[T] | triv_dst[T] | rot_kind[R] | triv_op[T]
    const size_t map_bytes = (2 * R + 3 * G + 2 * Tn) * 4 + R + Tn;
    MARK(0);
    int32_t rc = ensure_host_map(c, map_bytes);
    if (rc) return rc;
    MARK(1);
    int32_t *h_ra = (int32_t *)c->h_map, *h_rb = h_ra + R;
    int32_t *h_e0 = h_rb + R, *h_e1 = h_e0 + G, *h_dst = h_e1 + G, *h_ts = h_dst + G, *h_td = h_ts + Tn;
    uint8_t *h_kind = (uint8_t *)(h_td + Tn), *h_top = h_kind + R;
    {
        size_t r = 0, k = 0, t = 0;
        for (int64_t g = 0; g < B; g++) {
            const int op = opcodes[g];
            const int32_t ra = ia ? (op_has_a(op) ? ia[g] : 0) : (int32_t)g;
            const int32_t rb = ib ? (op_has_b(op) ? ib[g] : 0) : (int32_t)g;
            const int32_t rcw = ic ? (op == TFHE_GATE_MUX ? ic[g] : 0) : (int32_t)g;
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
    MARK(2);
    HIP_TRY(c, c->map.reserve(map_bytes));
    HIP_TRY(c, hipMemcpyAsync(c->map.p, c->h_map, map_bytes, hipMemcpyHostToDevice, s));
    MARK(3);
    HIP_TRY(c, hipEventRecord(c->map_ev, s));
    MARK(4);
    c->map_cur->pending = true;
    const int32_t *d_ra = (const int32_t *)c->map.p, *d_rb = d_ra + R;
    const int32_t *d_e0 = d_rb + R, *d_e1 = d_e0 + G, *d_dst = d_e1 + G, *d_ts = d_dst + G, *d_td = d_ts + Tn;
    const uint8_t *d_kind = (const uint8_t *)(d_td + Tn), *d_top = d_kind + R;

    const int n = c->P.n, kNn = c->P.k * c->P.N;
    const bool no_ev = !c->timing_events;       // option "timing_events": every event record costs the stream ~5 us between two kernels
    next_timing_slot(c);
    if (!no_ev) HIP_TRY(c, hipEventRecord(c->ev[0], s));
    if (R > 0) {
        HIP_TRY(c, c->bara.reserve(R * (size_t)(n + 1) * 4));
        HIP_TRY(c, c->ext.reserve(R * (size_t)(kNn + 1) * 4));
        hipLaunchKernelGGL(prologue_kernel, dim3((unsigned)R), dim3(256), 0, s, d_in0, d_in1, d_in2, d_ra, d_rb, d_kind,
                           (int32_t *)c->bara.p, n, ilog2i(2 * c->P.N));
        HIP_TRY(c, hipGetLastError());
    }
    MARK(5);
    if (!no_ev) HIP_TRY(c, hipEventRecord(c->ev[1], s));
    if (R > 0) {
        rc = launch_blind_rotate(c, R, (int32_t)(1u << 29), s);   // mu = encode_message(1, 8), gates.jl:17
        if (rc) return rc;
    } else {
        c->diag_rows = 0;
    }
    MARK(6);
    if (!no_ev) HIP_TRY(c, hipEventRecord(c->ev[2], s));
    if (G > 0) {
        rc = launch_keyswitch(c, G, d_e0, d_e1, d_dst, (const int32_t *)c->ext.p, d_out, s);
        if (rc) return rc;
    }
    if (!no_ev) HIP_TRY(c, hipEventRecord(c->ev[3], s));
    if (Tn > 0) {
        hipLaunchKernelGGL(trivial_gates_kernel, dim3((unsigned)Tn), dim3(256), 0, s, d_in0, d_ts, d_td, d_top, d_out, n);
        HIP_TRY(c, hipGetLastError());
    }
    if (!no_ev) commit_timing_slot(c);
    c->last_rotations = (int64_t)R;
    MARK(7);
    if (dbg_host) fprintf(stderr, "run_gates host us: classify+enter %.0f | stage %.0f | fill %.0f | map H2D %.0f | map event %.0f | prologue %.0f | blind rotate %.0f | keyswitch+trivial %.0f\n",
                          t_acc[0], t_acc[1], t_acc[2], t_acc[3], t_acc[4], t_acc[5], t_acc[6], t_acc[7]);
    return leave_stream(c, s);
}

int32_t tfhe_gates_batch_dev(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *d_in0, const int32_t *d_in1,
                             const int32_t *d_in2, int32_t *d_out, int64_t B, void *stream)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (c->multi()) {
        if (c->kids.size() != 1) return c->set_err(TFHE_ERR_STATE, "gates_batch_dev: device pointers belong to one device; use tfhe_gates_batch on a multi-device context");
        const int32_t rc = tfhe_gates_batch_dev(c->kids[0], opcodes, d_in0, d_in1, d_in2, d_out, B, stream);
        if (rc) c->err = c->kids[0]->err;
        c->kid_ran[0] = 1;
        return rc;
    }
    if (B < 0 || (B > 0 && (!opcodes || !d_out))) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: NULL argument or negative B");
    if (B == 0) { c->timing_valid = false; c->last_rotations = 0; return TFHE_OK; }
    if (B > (int64_t)1 << 30) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: B too large");
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "gates_batch: context is multi-key");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return run_gates(c, "gates_batch", opcodes, B, d_in0, d_in1, d_in2, d_out, nullptr, nullptr, nullptr, nullptr, s);
}

// ---- levelised circuit execution on a device-resident wire table -------------------------
// On a multi-device context every device holds a replica of the wire table; the context tracks which replicas hold each
// wire's current value (wire_valid / wire_owner) and pull_wires brings a device up to date, device to device, for exactly the
// rows it is about to read.
static int32_t pull_wires(tfhe_ctx *c, int dst, const int32_t *wires, int64_t count);

int32_t tfhe_wires_alloc(tfhe_ctx *c, int64_t num_wires)
{
    ENTER_CTX(c);
    if (c->multi()) {      // one replica of the table per device
        const int32_t rc = fan_out(c, all_kids(c), [&](int k) { return tfhe_wires_alloc(c->kids[(size_t)k], num_wires); });
        if (rc == TFHE_OK) {
            c->num_wires = num_wires;
            const size_t nw = (size_t)std::max<int64_t>(num_wires, 0);
            c->wire_valid.assign(c->kids.size(), std::vector<uint8_t>(nw, 1));
            c->wire_owner.assign(nw, 0);
        }
        return rc;
    }
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
    if (!c->multi() && !c->d_wires) return c->set_err(TFHE_ERR_STATE, "%s: no wire table allocated", who);
    if (c->multi() && c->num_wires <= 0) return c->set_err(TFHE_ERR_STATE, "%s: no wire table allocated", who);
    if (first < 0 || count < 0 || first + count > c->num_wires || (count > 0 && !host))
        return c->set_err(TFHE_ERR_INVALID_ARG, "%s: wire range [%lld, %lld) outside the table of %lld wires or NULL buffer", who,
                          (long long)first, (long long)(first + count), (long long)c->num_wires);
    return TFHE_OK;
}

int32_t tfhe_wires_upload(tfhe_ctx *c, int64_t first, int64_t count, const int32_t *host)
{
    ENTER_CTX(c);
    if (c->multi()) {      // every replica takes the rows: they are valid everywhere afterwards
        const int32_t rc0 = wires_range_ok(c, "wires_upload", first, count, host);
        if (rc0 || count == 0) return rc0;
        const int32_t rc = fan_out(c, all_kids(c), [&](int k) { return tfhe_wires_upload(c->kids[(size_t)k], first, count, host); });
        for (size_t k = 0; k < c->kids.size(); k++)      // (after a failed upload the range is current nowhere: the caller got the error and uploads again)
            std::fill(c->wire_valid[k].begin() + first, c->wire_valid[k].begin() + first + count, rc == TFHE_OK ? 1 : 0);
        if (rc == TFHE_OK) std::fill(c->wire_owner.begin() + first, c->wire_owner.begin() + first + count, 0);
        return rc;
    }
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
    ENTER_CTX(c);
    if (c->multi()) {      // the first device is brought up to date for the range, then read
        int32_t rc = wires_range_ok(c, "wires_download", first, count, host);
        if (rc || count == 0) return rc;
        std::vector<int32_t> idx((size_t)count);
        for (int64_t i = 0; i < count; i++) idx[(size_t)i] = (int32_t)(first + i);
        rc = pull_wires(c, 0, idx.data(), count);
        if (rc) return rc;
        rc = tfhe_wires_download(c->kids[0], first, count, host);
        if (rc) c->err = c->kids[0]->err;
        return rc;
    }
    int32_t rc = wires_range_ok(c, "wires_download", first, count, host);
    if (rc || count == 0) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t row = (size_t)(c->P.n + 1) * 4;
    HIP_TRY(c, hipMemcpyAsync(host, (const char *)c->d_wires + (size_t)first * row, (size_t)count * row, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TFHE_OK;
}


int32_t tfhe_wires_gather(tfhe_ctx *c, const int32_t *wires, int64_t count, int32_t *host)
{
    ENTER_CTX(c);
    if (c->multi()) {
        if (c->num_wires <= 0) return c->set_err(TFHE_ERR_STATE, "wires_gather: no wire table allocated");
        if (count < 0 || (count > 0 && (!wires || !host))) return c->set_err(TFHE_ERR_INVALID_ARG, "wires_gather: NULL argument or negative count");
        for (int64_t i = 0; i < count; i++)
            if (wires[i] < 0 || wires[i] >= c->num_wires) return c->set_err(TFHE_ERR_INVALID_ARG, "wires_gather: wire %d outside the table of %lld wires", wires[i], (long long)c->num_wires);
        int32_t rc = pull_wires(c, 0, wires, count);
        if (rc) return rc;
        rc = tfhe_wires_gather(c->kids[0], wires, count, host);
        if (rc) c->err = c->kids[0]->err;
        return rc;
    }
    if (!c->d_wires) return c->set_err(TFHE_ERR_STATE, "wires_gather: no wire table allocated");
    if (count < 0 || (count > 0 && (!wires || !host))) return c->set_err(TFHE_ERR_INVALID_ARG, "wires_gather: NULL argument or negative count");
    if (count == 0) return TFHE_OK;
    for (int64_t i = 0; i < count; i++)
        if (wires[i] < 0 || wires[i] >= c->num_wires) return c->set_err(TFHE_ERR_INVALID_ARG, "wires_gather: wire %d outside the table of %lld wires", wires[i], (long long)c->num_wires);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = c->stream;
    { const int32_t rc0 = enter_stream(c, s); if (rc0) return rc0; }
    const int n1 = c->P.n + 1;
    int32_t rc = ensure_host_map(c, (size_t)count * 4);
    if (rc) return rc;
    memcpy(c->h_map, wires, (size_t)count * 4);
    HIP_TRY(c, c->map.reserve((size_t)count * 4));
    HIP_TRY(c, hipMemcpyAsync(c->map.p, c->h_map, (size_t)count * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, c->io[3].reserve((size_t)count * n1 * 4));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)count), dim3(256), 0, s, (const int32_t *)c->d_wires, (const int32_t *)c->map.p, (int32_t *)c->io[3].p, n1);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(host, c->io[3].p, (size_t)count * n1 * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return TFHE_OK;
}

// every index in range; no wire both written and read inside one level (the level's gates are independent).
// O(B) work whatever the size of the wire table: only the level's own output wires are hashed.
static int32_t validate_level(tfhe_ctx *c, int64_t num_wires, const uint8_t *opcodes, const int32_t *a, const int32_t *b, const int32_t *cc,
                              const int32_t *out, int64_t B)
{
    std::unordered_set<int32_t> written;
    written.reserve((size_t)B * 2);
    auto bad = [&](int64_t v) { return v < 0 || v >= num_wires; };
    for (int64_t g = 0; g < B; g++) {
        const int op = opcodes[g];
        if (op >= TFHE_GATE__COUNT) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: bad opcode %d at gate %lld", op, (long long)g);
        const bool has_a = op_has_a(op), has_b = op_has_b(op), has_c = (op == TFHE_GATE_MUX);
        if ((has_a && (!a || bad(a[g]))) || (has_b && (!b || bad(b[g]))) || (has_c && (!cc || bad(cc[g]))) || bad(out[g]))
            return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: wire index out of range (or missing operand array) at gate %lld", (long long)g);
        if (!written.insert(out[g]).second) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: wire %d written twice in one level", out[g]);
    }
    for (int64_t g = 0; g < B; g++) {
        const int op = opcodes[g];
        if ((op_has_a(op) && written.count(a[g])) || (op_has_b(op) && written.count(b[g])) || (op == TFHE_GATE_MUX && written.count(cc[g])))
            return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: gate %lld reads a wire written in the same level", (long long)g);
    }
    return TFHE_OK;
}

// Brings device `dst`'s replica up to date for wires[0 .. count): every listed wire whose current value is elsewhere is fetched
// from its owner — per source device ONE gather on the source's stream, ONE copy, ONE scatter on the destination's stream,
// ordered by events; nothing is waited for on the host (except a ring slot still in flight from four transfers ago).
//   peer path  (level_exchange 0 where peer_ok, or 1):  gather -> hipMemcpyPeerAsync on the destination's stream -> scatter
//   host path  (no peer access, or level_exchange 2):   gather -> D2H into the pair's pinned block on the source's stream
//                                                       -> H2D on the destination's stream -> scatter
// Everything runs on the kids' own streams, so it is ordered with the levels before and after it on each device.
static int32_t pull_wires(tfhe_ctx *c, int dst, const int32_t *wires, int64_t count)
{
    const int nk = (int)c->kids.size();
    std::vector<std::vector<int32_t>> from((size_t)nk);
    std::vector<uint8_t> &valid = c->wire_valid[(size_t)dst];
    for (int64_t i = 0; i < count; i++) {
        const int32_t w = wires[i];
        if (w < 0 || valid[(size_t)w]) continue;
        valid[(size_t)w] = 1;                               // (also drops duplicates in the list)
        from[(size_t)c->wire_owner[(size_t)w]].push_back(w);
    }
    tfhe_ctx *kd = c->kids[(size_t)dst];
    const size_t row = (size_t)(c->P.n + 1) * 4;
    const int n1 = c->P.n + 1;
    int32_t rc = TFHE_OK;
    for (int src = 0; src < nk && rc == TFHE_OK; src++) {
        const std::vector<int32_t> &rows = from[(size_t)src];
        if (rows.empty()) continue;
        tfhe_ctx *ks = c->kids[(size_t)src];
        auto body = [&]() -> int32_t {
            PairXfer *&px = c->xfer[(size_t)src * nk + dst];
            if (!px) px = new PairXfer();
            PairXfer::Slot &sl = px->slot[px->next++ % PairXfer::kSlots];
            const size_t cnt = rows.size(), ibytes = cnt * 4, rbytes = cnt * row;
            const bool peer = c->level_exchange == 1 || (c->level_exchange == 0 && c->peer_ok[(size_t)src * nk + dst]);
            if (sl.pending) { HIP_TRY(c, hipEventSynchronize(sl.taken)); sl.pending = false; }
            // source side
            HIP_TRY(c, hipSetDevice(ks->device));
            if (!sl.ready) HIP_TRY(c, hipEventCreateWithFlags(&sl.ready, hipEventDisableTiming));
            if (ibytes > sl.h_idx_cap) {
                if (sl.h_idx) (void)hipHostFree(sl.h_idx);
                sl.h_idx = nullptr; sl.h_idx_cap = 0;
                HIP_TRY(c, hipHostMalloc((void **)&sl.h_idx, ibytes * 2 + 256, hipHostMallocPortable));
                sl.h_idx_cap = ibytes * 2 + 256;
            }
            memcpy(sl.h_idx, rows.data(), ibytes);
            HIP_TRY(c, sl.out.reserve(rbytes));
            HIP_TRY(c, sl.idx_src.reserve(ibytes));
            HIP_TRY(c, hipMemcpyAsync(sl.idx_src.p, sl.h_idx, ibytes, hipMemcpyHostToDevice, ks->stream));
            hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)cnt), dim3(256), 0, ks->stream, (const int32_t *)ks->d_wires, (const int32_t *)sl.idx_src.p, (int32_t *)sl.out.p, n1);
            HIP_TRY(c, hipGetLastError());
            if (!peer) {
                if (rbytes > sl.h_rows_cap) {
                    if (sl.h_rows) (void)hipHostFree(sl.h_rows);
                    sl.h_rows = nullptr; sl.h_rows_cap = 0;
                    HIP_TRY(c, hipHostMalloc(&sl.h_rows, rbytes + rbytes / 4 + 256, hipHostMallocPortable));
                    sl.h_rows_cap = rbytes + rbytes / 4 + 256;
                }
                HIP_TRY(c, hipMemcpyAsync(sl.h_rows, sl.out.p, rbytes, hipMemcpyDeviceToHost, ks->stream));
            }
            HIP_TRY(c, hipEventRecord(sl.ready, ks->stream));
            ks->own_pending = true;
            // destination side
            HIP_TRY(c, hipSetDevice(kd->device));
            if (!sl.taken) HIP_TRY(c, hipEventCreateWithFlags(&sl.taken, hipEventDisableTiming));
            HIP_TRY(c, sl.in.reserve(rbytes));
            HIP_TRY(c, sl.idx_dst.reserve(ibytes));
            HIP_TRY(c, hipMemcpyAsync(sl.idx_dst.p, sl.h_idx, ibytes, hipMemcpyHostToDevice, kd->stream));
            HIP_TRY(c, hipStreamWaitEvent(kd->stream, sl.ready, 0));
            if (peer) HIP_TRY(c, hipMemcpyPeerAsync(sl.in.p, kd->device, sl.out.p, ks->device, rbytes, kd->stream));
            else HIP_TRY(c, hipMemcpyAsync(sl.in.p, sl.h_rows, rbytes, hipMemcpyHostToDevice, kd->stream));
            hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)cnt), dim3(256), 0, kd->stream, (const int32_t *)sl.in.p, (const int32_t *)sl.idx_dst.p, kd->d_wires, n1);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipEventRecord(sl.taken, kd->stream));
            sl.pending = true;
            kd->own_pending = true;
            return TFHE_OK;
        };
        rc = body();
        if (rc)      // what was not fetched is not valid here (the rows of this and the later sources)
            for (int s2 = src; s2 < nk; s2++)
                for (int32_t w : from[(size_t)s2]) valid[(size_t)w] = 0;
    }
    return rc;
}

// One level on a multi-device context.  A level of fewer than level_split_min blind rotations runs on the first device only
// (a level costs one blind-rotation latency however few gates it has: spreading a narrow one buys nothing); a wide level —
// the 16 parallel MUXes of examples/tutorial.jl:60 are the small case — is cut into rotation-balanced contiguous shards, one
// per device.  Either way a device first fetches the operand rows whose current value is elsewhere (pull_wires), then runs
// its gates on its own stream; the outputs are valid on the device that computed them and travel only if somebody else reads
// them later.  The call returns when everything is queued, exactly as tfhe_gates_level on a one-device context does.
static int32_t multi_gates_level(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *a, const int32_t *b, const int32_t *cc,
                                 const int32_t *out, int64_t B)
{
    const int nk = (int)c->kids.size();
    int32_t rc = validate_level(c, c->num_wires, opcodes, a, b, cc, out, B);
    if (rc) return rc;
    int64_t R = 0;
    for (int64_t g = 0; g < B; g++) R += opcodes[g] == TFHE_GATE_MUX ? 2 : op_has_b(opcodes[g]) ? 1 : 0;
    std::fill(c->kid_ran.begin(), c->kid_ran.end(), 0);
    std::vector<int64_t> bounds((size_t)nk + 1, B);
    bounds[0] = 0;
    if (!(nk == 1 || c->level_split_min < 0 || R < c->level_split_min)) shard_bounds_by_rotations(opcodes, B, nk, bounds.data());
    auto off = [&](const int32_t *p, int64_t g) { return p ? p + g : nullptr; };
    // Phase 1: every shard's device fetches the operand rows it lacks.  All transfers are queued BEFORE any shard's gates: a
    // gather sits in its source device's stream, and queued behind that device's own shard of THIS level it would make the
    // reader wait for a level it does not depend on (a level reads only what earlier levels wrote).
    std::vector<int32_t> reads;
    for (int r = 0; r < nk; r++) {
        const int64_t s0 = bounds[(size_t)r], cnt = bounds[(size_t)r + 1] - s0;
        if (cnt <= 0) continue;
        reads.clear();
        for (int64_t g = s0; g < s0 + cnt; g++) {
            const int op = opcodes[g];
            if (op_has_a(op)) reads.push_back(a[g]);
            if (op_has_b(op)) reads.push_back(b[g]);
            if (op == TFHE_GATE_MUX) reads.push_back(cc[g]);
        }
        rc = pull_wires(c, r, reads.data(), (int64_t)reads.size());
        if (rc) return rc;
    }
    // Phase 2: the shards, each on its device's stream
    for (int r = 0; r < nk; r++) {
        const int64_t s0 = bounds[(size_t)r], cnt = bounds[(size_t)r + 1] - s0;
        if (cnt <= 0) continue;
        rc = tfhe_gates_level(c->kids[(size_t)r], opcodes + s0, off(a, s0), off(b, s0), off(cc, s0), out + s0, cnt);
        c->kid_ran[(size_t)r] = 1;
        // the shard's output rows are current on device r only (on failure nowhere that can be named; the caller got the error)
        for (int64_t g = s0; g < s0 + cnt; g++) {
            const size_t w = (size_t)out[g];
            for (int k = 0; k < nk; k++) c->wire_valid[(size_t)k][w] = (k == r);
            c->wire_owner[w] = r;
        }
        if (rc) return c->set_err(rc, "device %d (kid %d): %s", c->kids[(size_t)r]->device, r, c->kids[(size_t)r]->err.c_str());
    }
    return TFHE_OK;
}

int32_t tfhe_gates_level(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *a, const int32_t *b, const int32_t *cc,
                         const int32_t *out, int64_t B)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!opcodes || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_level: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "gates_level: context is multi-key");
    if (c->multi()) {
        if (c->num_wires <= 0) return c->set_err(TFHE_ERR_STATE, "gates_level: no wire table allocated");
        return multi_gates_level(c, opcodes, a, b, cc, out, B);
    }
    if (!c->d_wires) return c->set_err(TFHE_ERR_STATE, "gates_level: no wire table allocated");
    { const int32_t rcv = validate_level(c, c->num_wires, opcodes, a, b, cc, out, B); if (rcv) return rcv; }
    HIP_TRY(c, hipSetDevice(c->device));
    // run_gates substitutes row 0 for operands an opcode does not read; a NULL index array (never read, checked above)
    // is replaced by `out` only so that the pointer is non-NULL
    return run_gates(c, "gates_level", opcodes, B, c->d_wires, c->d_wires, c->d_wires, c->d_wires, a ? a : out, b ? b : out, cc ? cc : out, out,
                     c->stream);
}

static int32_t gates_batch_one_stream(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                      int32_t *out, int64_t B, const bool (&need)[3]);

// The twin of a one-device context: same device, its own stream / workspaces / events, the OWNER's keys (read-only on the
// device).  Made on first use and kept; a key reload synchronises it and clears its borrowed pointers (quiesce), and every
// use re-points it at the owner's current keys and copies the owner's tunables.
static int32_t ensure_twin(tfhe_ctx *c)
{
    if (!c->twin) {
        tfhe_ctx *t = nullptr;
        const int32_t rc = tfhe_ctx_create(&c->P, c->device, &t);
        if (rc) return c->set_err(rc, "gates_batch: second stream context: %s", g_create_error.c_str());
        t->borrows_keys = true;
        t->pipeline_min = -1;
        c->twin = t;
    }
    tfhe_ctx *t = c->twin;
    t->d_bk = c->d_bk; t->bk_polys = c->bk_polys; t->d_ks = c->d_ks; t->d_ksp = c->d_ksp; t->ks_stride = c->ks_stride;
    t->d_ks4 = c->d_ks4; t->ks4_wtiles = c->ks4_wtiles; t->ks_mode = c->ks_mode; t->have_bk = c->have_bk; t->have_ks = c->have_ks;
    t->ks_slices_large = c->ks_slices_large; t->ks_variant = c->ks_variant; t->br_small = c->br_small; t->br_prio_pct = c->br_prio_pct;
    t->br_tiny = c->br_tiny; t->br_rt_l = c->br_rt_l; t->timing_events = c->timing_events; t->br_split = c->br_split; t->br_general = c->br_general; t->n2048_rw = c->n2048_rw; t->v3_rw = c->v3_rw; t->k2_rw = c->k2_rw; t->w2_rw = c->w2_rw;
    t->br_anyn = c->br_anyn; t->anyn_spec = c->anyn_spec; t->k2_w3 = c->k2_w3; t->n512_rw = c->n512_rw; t->n512_w2 = c->n512_w2;
    return TFHE_OK;
}

// fan-out of a host-buffer batch call: kid r takes gates [bounds[r], bounds[r+1])
static int32_t multi_gates_batch(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                 int32_t *out, int64_t B)
{
    const int nk = (int)c->kids.size();
    for (int64_t g = 0; g < B; g++)
        if (opcodes[g] >= TFHE_GATE__COUNT) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: bad opcode %d at gate %lld", opcodes[g], (long long)g);
    std::vector<int64_t> bounds((size_t)nk + 1);
    shard_bounds_by_rotations(opcodes, B, nk, bounds.data());
    std::vector<int> which;
    for (int r = 0; r < nk; r++)
        if (bounds[(size_t)r + 1] > bounds[(size_t)r]) which.push_back(r);
    const size_t n1 = (size_t)c->P.n + 1;
    return fan_out(c, which, [&](int r) {
        const int64_t s0 = bounds[(size_t)r], cnt = bounds[(size_t)r + 1] - s0;
        auto off = [&](const int32_t *p) { return p ? p + (size_t)s0 * n1 : nullptr; };
        return tfhe_gates_batch(c->kids[(size_t)r], opcodes + s0, off(in0), off(in1), off(in2), out + (size_t)s0 * n1, cnt);
    });
}

int32_t tfhe_gates_batch(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1,
                         const int32_t *in2, int32_t *out, int64_t B)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!opcodes || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->multi()) return multi_gates_batch(c, opcodes, in0, in1, in2, out, B);
    HIP_TRY(c, hipSetDevice(c->device));
    // which operand arrays do the opcodes read at all?  (an array nobody reads is not uploaded)
    bool need[3] = {false, false, false};
    for (int64_t g = 0; g < B; g++) {
        const int op = opcodes[g];
        if (op >= TFHE_GATE__COUNT) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: bad opcode %d at gate %lld", op, (long long)g);
        need[0] = need[0] || op_has_a(op); need[1] = need[1] || op_has_b(op); need[2] = need[2] || op == TFHE_GATE_MUX;
    }
    if ((need[0] && !in0) || (need[1] && !in1) || (need[2] && !in2))
        return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch: an operand array required by the opcodes is NULL");
    if (c->pipeline_min >= 0 && B >= c->pipeline_min && B >= 2 && !c->borrows_keys && !c->measure_margin && c->P.parties == 1 &&
        c->have_bk && c->have_ks) {
        // two halves, two streams: [upload B | compute A] then [compute B | download A] overlap
        int64_t bounds[3];
        shard_bounds_by_rotations(opcodes, B, 2, bounds);
        // ... cut at a ROUND boundary of the one-wave kernel where there is one near the middle: two halves of 4950 rotations
        // are 2.4 rounds each (three launches' worth of partly filled rounds), 4096 + 5804 are 2 + 2.8 — config 3's 8192-gate
        // shard through host buffers: 32.2 ms with the balanced cut, 30.2 ms on one stream (profiles/r04/)
        if (c->P.N == kN && c->P.k == 1 && (c->P.bs_l == 2 || c->P.bs_l == 3) && !c->br_general && !c->anyn()) {
            const int64_t resident = 8 * (int64_t)c->cu_count;
            int64_t total = 0;
            for (int64_t g = 0; g < B; g++) total += opcodes[g] == TFHE_GATE_MUX ? 2 : op_has_b(opcodes[g]) ? 1 : 0;
            const int64_t target = (total / 2 + resident / 2) / resident * resident;       // multiple of a round nearest to the middle
            if (target > 0 && target < total && total % resident != 0) {
                int64_t cum = 0, g = 0;
                while (g < B && cum < target) { cum += opcodes[g] == TFHE_GATE_MUX ? 2 : op_has_b(opcodes[g]) ? 1 : 0; g++; }
                if (cum == target || cum == target + 1) bounds[1] = g;      // (a MUX may straddle the boundary by one rotation)
            }
        }
        if (bounds[1] > 0 && bounds[1] < B) {
            const int32_t rct = ensure_twin(c);
            if (rct) return rct;
            const size_t n1 = (size_t)c->P.n + 1;
            auto off = [&](const int32_t *p, int64_t g) { return p ? p + (size_t)g * n1 : nullptr; };
            const int64_t h = bounds[1];
            int32_t rc1 = TFHE_OK;
            // (std::thread's constructor throws std::system_error when the process may not start another thread — a thread or
            //  process limit: no exception may cross the C ABI, so the batch then simply runs on one stream)
            std::thread second;
            try {
                second = std::thread([&] { rc1 = gates_batch_one_stream(c->twin, opcodes + h, off(in0, h), off(in1, h), off(in2, h), out + (size_t)h * n1, B - h, need); });
            } catch (const std::exception &) {
                return gates_batch_one_stream(c, opcodes, in0, in1, in2, out, B, need);
            }
            const int32_t rc0 = gates_batch_one_stream(c, opcodes, in0, in1, in2, out, h, need);
            second.join();
            if (rc1) return c->set_err(rc1, "gates_batch (second half): %s", c->twin->err.c_str());
            if (rc0) return rc0;
            c->last_rotations += c->twin->last_rotations;
            c->last_call_two_streams = true;
            return TFHE_OK;
        }
    }
    return gates_batch_one_stream(c, opcodes, in0, in1, in2, out, B, need);
}

// one host-buffer batch on one context's stream: uploads, kernels and download enqueued, nothing waited for
static int32_t gates_batch_enqueue(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                   int32_t *out, int64_t B, const bool (&need)[3])
{
    HIP_TRY(c, hipSetDevice(c->device));
    // the staging buffers io[] are reused by this call: order the copies behind the previous call's kernels
    { const int32_t rc0 = enter_stream(c, c->stream); if (rc0) return rc0; }
    const size_t bytes = (size_t)B * (c->P.n + 1) * 4;
    const int32_t *hin[3] = {in0, in1, in2};
    int32_t *din[3] = {nullptr, nullptr, nullptr};
    for (int i = 0; i < 3; i++) {
        if (!hin[i] || !need[i]) continue;
        HIP_TRY(c, c->io[i].reserve(bytes));
        HIP_TRY(c, hipMemcpyAsync(c->io[i].p, hin[i], bytes, hipMemcpyHostToDevice, c->stream));
        din[i] = (int32_t *)c->io[i].p;
    }
    HIP_TRY(c, c->io[3].reserve(bytes));
    int32_t rc = tfhe_gates_batch_dev(c, opcodes, din[0], din[1], din[2], (int32_t *)c->io[3].p, B, nullptr);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, bytes, hipMemcpyDeviceToHost, c->stream));
    return TFHE_OK;
}
// ... and synchronised
static int32_t gates_batch_one_stream(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                      int32_t *out, int64_t B, const bool (&need)[3])
{
    const int32_t rc = gates_batch_enqueue(c, opcodes, in0, in1, in2, out, B, need);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TFHE_OK;
}

// Streaming form: batch i + 1 is submitted while batch i computes.  Two slots — the context's own stream and its twin's —
// taken in turn: the upload of one batch runs under the kernels of the other and its download under the next one's.
int32_t tfhe_gates_batch_submit(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1,
                                const int32_t *in2, int32_t *out, int64_t B, int32_t *ticket)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!ticket) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch_submit: NULL ticket");
    *ticket = 2;                                   // "complete": nothing to wait for
    if (B < 0 || (B > 0 && (!opcodes || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch_submit: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->multi() && c->P.parties == 1) {
        // every device takes its rotation-balanced shard as a submit of its own (enqueued from this thread, one device after
        // the other: nothing is waited for), so each device keeps two batches in flight exactly as a one-device context does
        const int nk = (int)c->kids.size();
        for (int64_t g = 0; g < B; g++)
            if (opcodes[g] >= TFHE_GATE__COUNT) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch_submit: bad opcode %d at gate %lld", opcodes[g], (long long)g);
        const int slot = (int)(c->submits & 1u);
        if (c->slot_busy[slot]) { const int32_t rcw = tfhe_gates_batch_wait(c, slot); if (rcw) return rcw; }
        std::vector<int64_t> bounds((size_t)nk + 1);
        shard_bounds_by_rotations(opcodes, B, nk, bounds.data());
        const size_t n1 = (size_t)c->P.n + 1;
        c->kid_tickets[slot].assign((size_t)nk, 2);
        std::fill(c->kid_ran.begin(), c->kid_ran.end(), 0);
        for (int r = 0; r < nk; r++) {
            const int64_t s0 = bounds[(size_t)r], cnt = bounds[(size_t)r + 1] - s0;
            if (cnt <= 0) continue;
            auto off = [&](const int32_t *p) { return p ? p + (size_t)s0 * n1 : nullptr; };
            const int32_t rc = tfhe_gates_batch_submit(c->kids[(size_t)r], opcodes + s0, off(in0), off(in1), off(in2), out + (size_t)s0 * n1, cnt, &c->kid_tickets[slot][(size_t)r]);
            c->kid_ran[(size_t)r] = 1;
            if (rc) {
                c->set_err(rc, "device %d (kid %d): %s", c->kids[(size_t)r]->device, r, c->kids[(size_t)r]->err.c_str());
                c->slot_busy[slot] = true;                  // what the earlier kids took is waited for before the error is returned
                (void)tfhe_gates_batch_wait(c, slot);
                return rc;
            }
        }
        c->submits++;
        c->slot_busy[slot] = true;
        *ticket = slot;
        return TFHE_OK;
    }
    // contexts that cannot run two batches side by side (multi-key, diagnostics, a borrowed key) run this one now
    if (c->multi() || c->P.parties != 1 || c->measure_margin || c->borrows_keys || !c->have_bk || !c->have_ks)
        return tfhe_gates_batch(c, opcodes, in0, in1, in2, out, B);
    HIP_TRY(c, hipSetDevice(c->device));
    bool need[3] = {false, false, false};
    for (int64_t g = 0; g < B; g++) {
        const int op = opcodes[g];
        if (op >= TFHE_GATE__COUNT) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch_submit: bad opcode %d at gate %lld", op, (long long)g);
        need[0] = need[0] || op_has_a(op); need[1] = need[1] || op_has_b(op); need[2] = need[2] || op == TFHE_GATE_MUX;
    }
    if ((need[0] && !in0) || (need[1] && !in1) || (need[2] && !in2))
        return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch_submit: an operand array required by the opcodes is NULL");
    const int slot = (int)(c->submits & 1u);
    if (slot == 1) { const int32_t rct = ensure_twin(c); if (rct) return rct; }
    tfhe_ctx *t = slot ? c->twin : c;
    if (c->slot_busy[slot]) {                      // a third batch: the oldest one's buffers and workspaces are about to be reused
        HIP_TRY(c, hipStreamSynchronize(t->stream));
        c->slot_busy[slot] = false;
    }
    const int32_t rc = gates_batch_enqueue(t, opcodes, in0, in1, in2, out, B, need);
    if (rc) return slot ? c->set_err(rc, "gates_batch_submit: %s", t->err.c_str()) : rc;
    if (slot) { c->last_rotations = t->last_rotations; c->last_kernel = t->last_kernel; }     // tfhe_last_rotation_count / _kernel_name describe the newest submit, whichever stream took it
    c->submits++;
    c->slot_busy[slot] = true;
    *ticket = slot;
    return TFHE_OK;
}

int32_t tfhe_gates_batch_wait(tfhe_ctx *c, int32_t ticket)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (ticket == 2) return TFHE_OK;
    if (ticket != 0 && ticket != 1) return c->set_err(TFHE_ERR_INVALID_ARG, "gates_batch_wait: ticket %d was not issued by tfhe_gates_batch_submit", ticket);
    if (!c->slot_busy[ticket]) return TFHE_OK;     // already waited for (or displaced by a later submit, which waited)
    if (c->multi()) {
        int32_t first_rc = TFHE_OK;
        for (size_t r = 0; r < c->kids.size() && r < c->kid_tickets[ticket].size(); r++) {
            const int32_t rc = tfhe_gates_batch_wait(c->kids[r], c->kid_tickets[ticket][r]);
            if (rc && !first_rc) { first_rc = rc; c->set_err(rc, "device %d (kid %d): %s", c->kids[r]->device, (int)r, c->kids[r]->err.c_str()); }
        }
        c->slot_busy[ticket] = false;
        return first_rc;
    }
    tfhe_ctx *t = ticket ? c->twin : c;
    if (!t) return c->set_err(TFHE_ERR_STATE, "gates_batch_wait: ticket 1 was never issued (no batch has used the second stream)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(t->stream));
    c->slot_busy[ticket] = false;
    return TFHE_OK;
}

int32_t tfhe_bootstrap_batch(tfhe_ctx *c, int32_t mu, const int32_t *in, int32_t *out, int64_t B, int32_t with_keyswitch)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!in || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "bootstrap_batch: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "bootstrap_batch: context is multi-key");
    if (c->multi()) {
        const size_t wi = (size_t)c->P.n + 1, wo = with_keyswitch ? wi : (size_t)c->P.k * c->P.N + 1;
        return multi_rows(c, B, [&](tfhe_ctx *k, int64_t s0, int64_t cnt) { return tfhe_bootstrap_batch(k, mu, in + (size_t)s0 * wi, out + (size_t)s0 * wo, cnt, with_keyswitch); });
    }
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
    next_timing_slot(c);
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
    rc = leave_stream(c, s);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(s));
    commit_timing_slot(c);
    c->last_rotations = B;
    return TFHE_OK;
}

int32_t tfhe_keyswitch_batch(tfhe_ctx *c, const int32_t *in, int32_t *out, int64_t B)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (B < 0 || (B > 0 && (!in || !out))) return c->set_err(TFHE_ERR_INVALID_ARG, "keyswitch_batch: NULL argument or negative B");
    if (B == 0) return TFHE_OK;
    if (c->P.parties != 1) return c->set_err(TFHE_ERR_STATE, "keyswitch_batch: context is multi-key");
    if (c->multi()) {
        const size_t wi = (size_t)c->P.k * c->P.N + 1, wo = (size_t)c->P.n + 1;
        return multi_rows(c, B, [&](tfhe_ctx *k, int64_t s0, int64_t cnt) { return tfhe_keyswitch_batch(k, in + (size_t)s0 * wi, out + (size_t)s0 * wo, cnt); });
    }
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
    next_timing_slot(c);
    HIP_TRY(c, hipEventRecord(c->ev[0], s));
    HIP_TRY(c, hipEventRecord(c->ev[1], s));
    HIP_TRY(c, hipEventRecord(c->ev[2], s));
    rc = launch_keyswitch(c, (size_t)B, (const int32_t *)c->map.p, nullptr, nullptr, (const int32_t *)c->ext.p, (int32_t *)c->io[3].p, s);
    if (rc) return rc;
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, out_bytes, hipMemcpyDeviceToHost, s));
    rc = leave_stream(c, s);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(s));
    commit_timing_slot(c);
    c->last_rotations = 0;
    c->diag_rows = 0;
    return TFHE_OK;
}

// key preparation for whichever kernel family serves this context: Int32 polynomials -> spectra (`scale` folded in), on `s`
static int32_t launch_bk_prepare(tfhe_ctx *c, const int32_t *d_polys, cplx *d_out, size_t npolys, double scale_if_tuned, bool key_scale, hipStream_t s)
{
    if (npolys == 0) return TFHE_OK;
    if (c->anyn()) {
        const int log2N = ilog2i(c->P.N), M = c->P.N / 2;
        const size_t ldsp = (size_t)anyn::padded_len(M > 0 ? M : 1) * sizeof(cplx);
        if (ldsp > 64 * 1024) LDS_TRY(c, ldsp, anyn::bk_prepare_kernel);
        hipLaunchKernelGGL(anyn::bk_prepare_kernel, dim3((unsigned)npolys), dim3((unsigned)anyn::threads_for(c->P.N)), ldsp, s, d_polys, d_out,
                           (const cplx *)c->d_anyn_tab, (const cplx *)(c->d_anyn_tab + M), log2N, key_scale ? 1.0 / (double)M : 1.0);
    } else {
        hipLaunchKernelGGL(bk_prepare_kernel, dim3((unsigned)npolys), dim3(64), 0, s, d_polys, d_out, c->T, scale_if_tuned);
    }
    HIP_TRY(c, hipGetLastError());
    return TFHE_OK;
}

static int32_t mk_load_bk_common(tfhe_ctx *c, const void *bk, int32_t parties, bool is_c128)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!bk) return c->set_err(TFHE_ERR_INVALID_ARG, "mk_load_bootstrap_key: NULL key pointer");
    if (parties < 2 || c->P.parties < parties)
        return c->set_err(TFHE_ERR_INVALID_ARG, "mk_load_bootstrap_key: parties must be at least 2 and not exceed the context's max_parties (mk_api.jl:94)");
    if (c->multi()) return fan_out(c, all_kids(c), [&](int k) { return mk_load_bk_common(c->kids[(size_t)k], bk, parties, is_c128); });
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->P.N, M = N / 2;
    const size_t per = (size_t)2 * c->P.bs_l * parties + 2 * c->P.bs_l;
    const size_t npolys = (size_t)parties * c->P.n * per;
    const size_t bytes_in = is_c128 ? npolys * M * sizeof(cplx) : npolys * N * 4;
    quiesce(c);
    if (c->d_mk_bk) { (void)hipFree(c->d_mk_bk); c->d_mk_bk = nullptr; c->have_mk_bk = false; }
    HIP_TRY(c, hipMalloc((void **)&c->d_mk_bk, npolys * M * sizeof(cplx)));
    void *d_in = nullptr;
    HIP_TRY(c, hipMalloc(&d_in, bytes_in));
    auto body = [&]() -> int32_t {
        HIP_TRY(c, hipMemcpyAsync(d_in, bk, bytes_in, hipMemcpyHostToDevice, c->stream));
        if (is_c128 && c->anyn()) {
            const size_t total = npolys * M;
            hipLaunchKernelGGL(anyn::bk_permute_c128_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, (const cplx *)d_in, c->d_mk_bk, ilog2i(c->P.N) - 1, total);
            HIP_TRY(c, hipGetLastError());
        } else if (is_c128) {
            hipLaunchKernelGGL(bk_permute_c128_kernel, dim3((unsigned)npolys), dim3(64), 0, c->stream, (const cplx *)d_in, c->d_mk_bk);
            HIP_TRY(c, hipGetLastError());
        } else {
            const int32_t rcp = launch_bk_prepare(c, (const int32_t *)d_in, c->d_mk_bk, npolys, 1.0 / kM, true, c->stream);
            if (rcp) return rcp;
        }
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return TFHE_OK;
    };
    const int32_t rc = body();
    (void)hipFree(d_in);
    if (rc) return rc;
    c->mk_parties = parties;
    c->have_mk_bk = true;
    return TFHE_OK;
}

int32_t tfhe_mk_load_bootstrap_key_i32(tfhe_ctx *c, const int32_t *bk, int32_t parties) { return mk_load_bk_common(c, bk, parties, false); }

int32_t tfhe_mk_load_bootstrap_key_c128(tfhe_ctx *c, const double *bk_spectra, int32_t parties) { return mk_load_bk_common(c, bk_spectra, parties, true); }

// RGSW.Expand on the device (mk_internals.jl:304-345, MKBootstrapKey :442-461): the parties' uni-encryptions and public
// keys in, the expanded transformed bootstrapping key resident on the device out.
int32_t tfhe_mk_expand_load_bootstrap_key(tfhe_ctx *c, int32_t parties, const int32_t *pub_b, const int32_t *c0, const int32_t *c1,
                                          const int32_t *d0, const int32_t *d1, const int32_t *f0, const int32_t *f1, int32_t *expanded_out)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!pub_b || !c0 || !c1 || !d0 || !d1 || !f0 || !f1) return c->set_err(TFHE_ERR_INVALID_ARG, "mk_expand: NULL argument");
    if (parties < 2 || c->P.parties < parties)
        return c->set_err(TFHE_ERR_INVALID_ARG, "mk_expand: parties must be at least 2 and not exceed the context's max_parties (mk_api.jl:94)");
    if (c->multi()) {
        // (the expanded key is written to the caller's buffer by the first device only)
        return fan_out(c, all_kids(c), [&](int k) {
            return tfhe_mk_expand_load_bootstrap_key(c->kids[(size_t)k], parties, pub_b, c0, c1, d0, d1, f0, f1, k == 0 ? expanded_out : nullptr);
        });
    }
    HIP_TRY(c, hipSetDevice(c->device));
    const int n = c->P.n, l = c->P.bs_l, P = parties;
    const size_t N = (size_t)c->P.N, M = N / 2;
    const bool any = c->anyn();
    const size_t per = (size_t)2 * l * P + 2 * l;
    const size_t npolys = (size_t)P * n * per;
    const size_t nl = (size_t)n * l;                   // polys per party in each of c0 .. f1
    c->have_mk_bk = false;
    quiesce(c);
    if (c->d_mk_bk) { (void)hipFree(c->d_mk_bk); c->d_mk_bk = nullptr; }
    HIP_TRY(c, hipMalloc((void **)&c->d_mk_bk, npolys * M * sizeof(cplx)));
    // scratch: the party's 6 uni-encryption arrays, the digit polynomials and their spectra, f0 / f1 spectra, the party's key slice
    int32_t *d_in = nullptr, *d_dec = nullptr, *d_key = nullptr;
    cplx *d_decs = nullptr, *d_fs = nullptr;
    const size_t ndec = (size_t)(P - 1) * l * l;
    auto cleanup = [&]() {
        if (d_in) (void)hipFree(d_in);
        if (d_dec) (void)hipFree(d_dec);
        if (d_key) (void)hipFree(d_key);
        if (d_decs) (void)hipFree(d_decs);
        if (d_fs) (void)hipFree(d_fs);
    };
    auto body = [&]() -> int32_t {
        HIP_TRY(c, hipMalloc((void **)&d_in, 6 * nl * N * 4));
        HIP_TRY(c, hipMalloc((void **)&d_dec, ndec * N * 4));
        HIP_TRY(c, hipMalloc((void **)&d_decs, ndec * M * sizeof(cplx)));
        HIP_TRY(c, hipMalloc((void **)&d_fs, 2 * nl * M * sizeof(cplx)));
        HIP_TRY(c, hipMalloc((void **)&d_key, (size_t)n * per * N * 4));
        std::vector<int32_t> h_dec(ndec * N);
        hipStream_t s = c->stream;
        for (int i = 0; i < P; i++) {
            // g^-1(b_q[jj] - b_i[jj])[u] for every other party q (tgsw.jl:99-117): [oq][u][jj][N]
            int oq = 0;
            for (int q = 0; q < P; q++) {
                if (q == i) continue;
                for (int jj = 0; jj < l; jj++) {
                    const int32_t *bq = pub_b + ((size_t)q * l + jj) * N, *bi = pub_b + ((size_t)i * l + jj) * N;
                    for (size_t t = 0; t < N; t++) {
                        const int32_t v = (int32_t)((uint32_t)bq[t] - (uint32_t)bi[t] + (uint32_t)c->g.offset);
                        for (int u = 0; u < l; u++) h_dec[(((size_t)oq * l + u) * l + jj) * N + t] = gadget_digit(v, u + 1, c->g);
                    }
                }
                oq++;
            }
            const int32_t *src[6] = {c0, c1, d0, d1, f0, f1};
            for (int a = 0; a < 6; a++)
                HIP_TRY(c, hipMemcpyAsync(d_in + (size_t)a * nl * N, src[a] + (size_t)i * nl * N, nl * N * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(d_dec, h_dec.data(), ndec * N * 4, hipMemcpyHostToDevice, s));
            const int32_t *dc0 = d_in, *dc1 = d_in + nl * N, *dd0 = d_in + 2 * nl * N, *dd1 = d_in + 3 * nl * N, *df = d_in + 4 * nl * N;
            int32_t rcp = launch_bk_prepare(c, (const int32_t *)d_dec, d_decs, ndec, 1.0, false, s);        // multiplier polynomials: unscaled
            if (rcp) return rcp;
            rcp = launch_bk_prepare(c, df, d_fs, 2 * nl, 1.0 / kM, true, s);                              // f0 then f1
            if (rcp) return rcp;
            if (any) {
                anyn::MkExpandArgs A;
                A.dec = d_decs; A.f = d_fs; A.d0 = dd0; A.key = d_key; A.wtab = c->d_anyn_tab; A.twist = c->d_anyn_tab + M;
                A.n = n; A.l = l; A.parties = P; A.party = i; A.log2N = ilog2i(c->P.N);
                const size_t ldsp = (size_t)anyn::padded_len(M > 0 ? (int)M : 1) * sizeof(cplx);
                if (ldsp > 64 * 1024) LDS_TRY(c, ldsp, anyn::mk_expand_kernel);
                hipLaunchKernelGGL(anyn::mk_expand_kernel, dim3((unsigned)n, (unsigned)(l * (P - 1)), 2), dim3((unsigned)anyn::threads_for(c->P.N)), ldsp, s, A);
            } else {
                MkExpandArgs A;
                A.dec = d_decs; A.f = d_fs; A.d0 = dd0; A.key = d_key; A.T = c->T; A.n = n; A.l = l; A.parties = P; A.party = i;
                hipLaunchKernelGGL(mk_expand_kernel, dim3((unsigned)n, (unsigned)(l * (P - 1)), 2), dim3(64), 0, s, A);
            }
            hipLaunchKernelGGL(mk_expand_copy_kernel, dim3((unsigned)n, (unsigned)l, 4), dim3(256), 0, s, dc0, dc1, dd0, dd1, d_key, n, l, P, i, (int)N);
            HIP_TRY(c, hipGetLastError());
            rcp = launch_bk_prepare(c, (const int32_t *)d_key, c->d_mk_bk + (size_t)i * n * per * M, (size_t)n * per, 1.0 / kM, true, s);
            if (rcp) return rcp;
            if (expanded_out)
                HIP_TRY(c, hipMemcpyAsync(expanded_out + (size_t)i * n * per * N, d_key, (size_t)n * per * N * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(c, hipStreamSynchronize(s));      // h_dec and the scratch buffers are reused by the next party
        }
        return TFHE_OK;
    };
    const int32_t rc = body();
    cleanup();
    if (rc) return rc;
    c->mk_parties = parties;
    c->have_mk_bk = true;
    return TFHE_OK;
}

int32_t tfhe_mk_load_keyswitch_key(tfhe_ctx *c, const int32_t *ks, int32_t parties)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!ks) return c->set_err(TFHE_ERR_INVALID_ARG, "mk_load_keyswitch_key: NULL key pointer");
    if (parties < 2 || c->P.parties < parties)
        return c->set_err(TFHE_ERR_INVALID_ARG, "mk_load_keyswitch_key: parties must be at least 2 and not exceed the context's max_parties");
    if (c->multi()) return fan_out(c, all_kids(c), [&](int k) { return tfhe_mk_load_keyswitch_key(c->kids[(size_t)k], ks, parties); });
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n1 = (size_t)c->P.n + 1, stride = (n1 + 3) & ~(size_t)3;
    const size_t rows = (size_t)c->P.N * c->P.ks_t * ((1u << c->P.ks_log2_base) - 1);   // per party (k = 1)
    c->have_mk_ks = false;
    quiesce(c);
    if (c->d_mk_ksp) { (void)hipFree(c->d_mk_ksp); c->d_mk_ksp = nullptr; }
    if (c->d_mk_ks4) { (void)hipFree(c->d_mk_ks4); c->d_mk_ks4 = nullptr; }
    if (c->d_ks) { (void)hipFree(c->d_ks); c->d_ks = nullptr; }
    // the kernel family by keyswitch shape, as for a single key (pick_ks_mode): int8 MFMA for base 4 / t = 8, the tiled integer
    // kernel for base 4 / t a multiple of 4, the gather kernel for every other base and length (keyswitch.jl:45-80 takes any)
    const bool ok3 = c->P.ks_log2_base == 2 && c->P.ks_t % 4 == 0 && c->P.N % KS3_SLICES == 0 && c->P.N / KS3_SLICES <= 128;
    const bool ok4 = c->P.ks_log2_base == 2 && c->P.ks_t == 8 && c->P.N % 128 == 0;
    const int mode = (c->ks_variant == 4 && ok4) ? 4 : (c->ks_variant >= 3 && ok3) ? 3 : 1;
    if (mode == 1) {
        // canonical layout, the parties' keys back to back
        HIP_TRY(c, hipMalloc((void **)&c->d_ks, (size_t)parties * rows * n1 * 4));
        HIP_TRY(c, hipMemcpyAsync(c->d_ks, ks, (size_t)parties * rows * n1 * 4, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->mk_ksp_words = rows * n1;
    } else
    if (mode == 3) {
        // (every copy and kernel of a loader runs on the context's own stream: nothing here depends on what the NULL stream orders)
        HIP_TRY(c, hipMalloc((void **)&c->d_mk_ksp, (size_t)parties * rows * stride * 4));
        HIP_TRY(c, hipMemsetAsync(c->d_mk_ksp, 0, (size_t)parties * rows * stride * 4, c->stream));
        HIP_TRY(c, hipMemcpy2DAsync(c->d_mk_ksp, stride * 4, ks, n1 * 4, n1 * 4, (size_t)parties * rows, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->mk_ksp_words = rows * stride;
        c->ks_stride = (int)stride;
    } else {   // MFMA fragments per party (keyswitch_kernel_v4)
        const int wtiles = (c->P.n + 1 + 31) / 32;
        const size_t frags = (size_t)c->P.N * wtiles * 4 * 64, words = rows * n1;
        int32_t *d_tmp = nullptr;
        HIP_TRY(c, hipMalloc((void **)&d_tmp, words * 4));
        auto body = [&]() -> int32_t {
            HIP_TRY(c, hipMalloc(&c->d_mk_ks4, (size_t)parties * frags * 16));
            for (int p = 0; p < parties; p++) {
                HIP_TRY(c, hipMemcpyAsync(d_tmp, ks + (size_t)p * words, words * 4, hipMemcpyHostToDevice, c->stream));
                hipLaunchKernelGGL(ks4_prepare_kernel, dim3((unsigned)((frags + 255) / 256)), dim3(256), 0, c->stream, (const int32_t *)d_tmp,
                                   (i32x4 *)c->d_mk_ks4 + (size_t)p * frags, c->P.n, c->P.N, wtiles);
                HIP_TRY(c, hipGetLastError());
                HIP_TRY(c, hipStreamSynchronize(c->stream));
            }
            return TFHE_OK;
        };
        const int32_t rc = body();
        (void)hipFree(d_tmp);        // also on the error path
        if (rc) return rc;
        c->mk_ks4_frags = frags;
        c->ks4_wtiles = wtiles;
    }
    c->ks_mode = mode;
    c->have_mk_ks = true;
    return TFHE_OK;
}

int32_t tfhe_mk_gate_nand_batch(tfhe_ctx *c, const int32_t *in0, const int32_t *in1, int32_t *out, int64_t B)
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
    hipLaunchKernelGGL(prologue_kernel, dim3((unsigned)B), dim3(256), 0, s, (const int32_t *)c->io[0].p, (const int32_t *)c->io[1].p,
                       (const int32_t *)nullptr, d_gate, d_gate, d_kind, (int32_t *)c->bara.p, NP * n, ilog2i(2 * c->P.N));
    HIP_TRY(c, hipGetLastError());
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
        const size_t lds2 = (size_t)rw * ((NP + 1) * kImg * 4 + (2 * kXchElems + kM) * sizeof(cplx)) + 64 * sizeof(cplx);
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
    // mk_keyswitch (mk_internals.jl:397-411): per party a single-key keyswitch of its mask column with b = 0
    if (c->ks_mode == 4) {
        Ks4Args a4;
        a4.ext = (const int32_t *)c->ext.p; a4.e0 = d_gate; a4.e1 = nullptr; a4.dst = nullptr; a4.out = (int32_t *)c->io[3].p;
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
        // any base / length: the gather kernel per party, the b words chained through the output (stream-ordered)
        KsArgs k1;
        k1.ext = (const int32_t *)c->ext.p; k1.e0 = d_gate; k1.e1 = nullptr; k1.dst = nullptr; k1.out = (int32_t *)c->io[3].p;
        k1.n = n; k1.kN = c->P.N; k1.t = c->P.ks_t; k1.log2_base = c->P.ks_log2_base;
        k1.in_stride = ew; k1.in_b = NP * c->P.N; k1.out_stride = nw; k1.out_b = NP * n;
        for (int p = 0; p < NP; p++) {
            k1.in_off = p * c->P.N; k1.out_off = p * n; k1.add_b = (p == 0);
            k1.ks = c->d_ks + (size_t)p * c->mk_ksp_words;
            hipLaunchKernelGGL(keyswitch_kernel, dim3((unsigned)B, (unsigned)((n + 1 + 256 * KS1_WPT - 1) / (256 * KS1_WPT))), dim3(256), 0, s, k1);
        }
    } else {
        Ks3Args k3;
        k3.ext = (const int32_t *)c->ext.p; k3.e0 = d_gate; k3.e1 = nullptr; k3.dst = nullptr; k3.out = (int32_t *)c->io[3].p;
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
    HIP_TRY(c, hipEventRecord(c->ev[3], s));
    HIP_TRY(c, hipMemcpyAsync(out, c->io[3].p, bytes, hipMemcpyDeviceToHost, s));
    rc = leave_stream(c, s);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(s));
    commit_timing_slot(c);
    c->last_rotations = B;
    return TFHE_OK;
}

int32_t tfhe_last_timing_ms(tfhe_ctx *c, int32_t which, float *ms)
{
    ENTER_CTX(c);
    if (!c || !ms) return TFHE_ERR_INVALID_ARG;
    if (c->multi()) {      // the shards ran concurrently: the batch took as long as the slowest of them
        float worst = -1.f;
        for (size_t k = 0; k < c->kids.size(); k++) {
            if (!c->kid_ran[k]) continue;
            float v = 0.f;
            const int32_t rc = tfhe_last_timing_ms(c->kids[k], which, &v);
            if (rc) { c->err = c->kids[k]->err; return rc; }
            worst = v > worst ? v : worst;
        }
        if (worst < 0.f) return c->set_err(TFHE_ERR_STATE, "last_timing: no batch call recorded");
        *ms = worst;
        return TFHE_OK;
    }
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
    if (c->last_call_two_streams && c->twin && c->twin->timing_valid) {
        // a two-stream call: from this (the first-started) stream's start of the phase to the LATER of the two streams' ends
        float other = 0.f;
        HIP_TRY(c, hipEventSynchronize(c->twin->ev[3]));
        HIP_TRY(c, hipEventElapsedTime(&other, c->ev[a], c->twin->ev[b]));
        if (other > *ms) *ms = other;
    }
    return TFHE_OK;
}

int32_t tfhe_timing_history_ms(tfhe_ctx *c, int32_t which, float *ms, int32_t max_calls, int32_t *n_out)
{
    ENTER_CTX(c);
    if (!c || !ms || !n_out || max_calls < 0) return TFHE_ERR_INVALID_ARG;
    *n_out = 0;
    if (c->multi()) return c->set_err(TFHE_ERR_STATE, "timing_history: ask a one-device context (a multi-device context reports the slowest shard per call: tfhe_last_timing_ms)");
    int a, b;
    switch (which) {
    case 0: a = 1; b = 2; break;
    case 1: a = 2; b = 3; break;
    case 2: a = 0; b = 3; break;
    default: return c->set_err(TFHE_ERR_INVALID_ARG, "timing_history: which must be 0, 1 or 2");
    }
    // (one slot fewer than the ring holds: the slot the NEXT call records into is the oldest committed set once the ring is full,
    //  and a call that fails part-way leaves it half overwritten — it is never part of the history)
    const int64_t have = std::min<int64_t>(c->timed_calls, tfhe_ctx::kTimingSlots - 1);
    const int32_t n = (int32_t)std::min<int64_t>(have, max_calls);
    if (n == 0) return TFHE_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventSynchronize(c->evring[(c->timed_calls - 1) % tfhe_ctx::kTimingSlots][3]));   // the newest complete call has finished: so have the older ones on this context
    for (int32_t i = 0; i < n; i++) {                   // oldest of the n first
        hipEvent_t *set = c->evring[(c->timed_calls - n + i) % tfhe_ctx::kTimingSlots];
        HIP_TRY(c, hipEventElapsedTime(ms + i, set[a], set[b]));
    }
    *n_out = n;
    return TFHE_OK;
}

int32_t tfhe_last_device_count(const tfhe_ctx *c)
{
    if (!c) return -1;
    if (!c->multi()) return 1;
    int32_t n = 0;
    for (size_t k = 0; k < c->kids.size(); k++) n += c->kid_ran[k] ? 1 : 0;
    return n;
}

int64_t tfhe_last_rotation_count(const tfhe_ctx *c)
{
    if (!c) return -1;
    if (c->multi()) {
        int64_t sum = 0;
        for (size_t k = 0; k < c->kids.size(); k++) if (c->kid_ran[k]) sum += c->kids[k]->last_rotations;
        return sum;
    }
    return c->last_rotations;
}

const char *tfhe_last_kernel_name(const tfhe_ctx *c)
{
    if (!c) return "";
    if (c->multi()) {
        for (size_t k = 0; k < c->kids.size(); k++) if (c->kid_ran[k]) return c->kids[k]->last_kernel.c_str();
        return "";
    }
    return c->last_kernel.c_str();
}

// reads the DIAG records of the last batch call: worst rounding margin and the median in-kernel clock
static int32_t read_diag(tfhe_ctx *c, double *worst, double *mhz)
{
    if (!c->diag_rows) return c->set_err(TFHE_ERR_STATE, "diagnostics: enable tfhe_set_option(\"measure_margin\", 1) before the batch call");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->done_pending) HIP_TRY(c, hipEventSynchronize(c->done_ev));
    const size_t R = c->diag_rows;
    std::vector<unsigned long long> h(3 * R);
    HIP_TRY(c, hipMemcpyAsync(h.data(), c->diag.p, 3 * R * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double m = 0;
    for (size_t r = 0; r < R; r++) {
        double v;
        memcpy(&v, &h[r], 8);
        m = v > m ? v : m;
    }
    if (worst) *worst = m;
    if (mhz) {
        std::vector<double> f;
        f.reserve(R);
        // s_memrealtime ticks at 100 MHz: a workgroup that lived less than 10 us (1000 ticks) gives a ratio with a percent-level
        // quantisation error and, right after the device woke up, sits on the DVFS ramp — such records are not a clock reading
        constexpr unsigned long long kMinTicks = 1000;
        for (size_t r = 0; r < R; r++)
            if (h[R + 2 * r + 1] >= kMinTicks) f.push_back((double)h[R + 2 * r] / (double)h[R + 2 * r + 1] * 100.0);
        if (f.empty()) return c->set_err(TFHE_ERR_STATE, "diagnostics: the kernel's workgroups ran for less than 10 us each: too short for a clock reading");
        if (getenv("TFHE_DEBUG_LIFETIMES")) {       // development aid: wave lifetimes from the 100 MHz s_memrealtime counter
            double sum = 0, mn = 1e30, mx = 0;
            for (size_t r = 0; r < R; r++) {
                const double us = (double)h[R + 2 * r + 1] * 0.01;
                sum += us; mn = us < mn ? us : mn; mx = us > mx ? us : mx;
            }
            fprintf(stderr, "lifetimes of %zu rotations: mean %.1f us, min %.1f, max %.1f\n", R, sum / (double)R, mn, mx);
            double byx[8] = {0}; size_t nx[8] = {0};
            for (size_t r = 0; r < R; r++) { byx[r & 7] += (double)h[R + 2 * r + 1] * 0.01; nx[r & 7]++; }
            fprintf(stderr, "  mean by (rotation mod 8):");
            for (int x = 0; x < 8; x++) fprintf(stderr, " %.0f", nx[x] ? byx[x] / (double)nx[x] : 0.0);
            fprintf(stderr, "\n  mean by block of R/8 consecutive rotations:");
            for (int b = 0; b < 8; b++) {
                double sb = 0; size_t nb = 0;
                for (size_t r = b * (R / 8); r < (b + 1) * (R / 8); r++) { sb += (double)h[R + 2 * r + 1] * 0.01; nb++; }
                fprintf(stderr, " %.0f", nb ? sb / (double)nb : 0.0);
            }
            fprintf(stderr, "\n  first 32 lifetimes:");
            for (size_t r = 0; r < 32 && r < R; r++) fprintf(stderr, " %.0f", (double)h[R + 2 * r + 1] * 0.01);
            fprintf(stderr, "\n");
        }
        std::nth_element(f.begin(), f.begin() + (long)(f.size() / 2), f.end());
        *mhz = f[f.size() / 2];
    }
    return TFHE_OK;
}

int32_t tfhe_last_rounding_margin(tfhe_ctx *c, double *worst)
{
    ENTER_CTX(c);
    if (!c || !worst) return TFHE_ERR_INVALID_ARG;
    if (c->multi()) {
        double m = -1;
        for (size_t k = 0; k < c->kids.size(); k++) {
            if (!c->kid_ran[k]) continue;
            double v = 0;
            const int32_t rc = read_diag(c->kids[k], &v, nullptr);
            if (rc) { c->err = c->kids[k]->err; return rc; }
            m = v > m ? v : m;
        }
        if (m < 0) return c->set_err(TFHE_ERR_STATE, "last_rounding_margin: no batch call recorded");
        *worst = m;
        return TFHE_OK;
    }
    return read_diag(c, worst, nullptr);
}

int32_t tfhe_last_kernel_clock_mhz(tfhe_ctx *c, double *mhz)
{
    ENTER_CTX(c);
    if (!c || !mhz) return TFHE_ERR_INVALID_ARG;
    if (c->multi()) {
        for (size_t k = 0; k < c->kids.size(); k++) {
            if (!c->kid_ran[k]) continue;
            const int32_t rc = read_diag(c->kids[k], nullptr, mhz);
            if (rc) c->err = c->kids[k]->err;
            return rc;
        }
        return c->set_err(TFHE_ERR_STATE, "last_kernel_clock: no batch call recorded");
    }
    return read_diag(c, nullptr, mhz);
}

#ifdef TFHE_STAMP
// development builds only (make stamp): per-phase shader-clock ticks of workgroup 0's waves, [4 waves][16]
int32_t tfhe_debug_phases(tfhe_ctx *c, unsigned long long *out64)
{
    if (!c || !out64 || c->multi() || !c->diag_rows) return TFHE_ERR_STATE;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpyAsync(out64, (const char *)c->diag.p + c->diag_rows * 24, 64 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return TFHE_OK;
}
#endif

// the current value of an option tfhe_set_option takes (so that a caller that changes one for a while can put it back)
int32_t tfhe_get_option(tfhe_ctx *c, const char *name, int64_t *value)
{
    ENTER_CTX(c);
    if (!name || !value) return c->set_err(TFHE_ERR_INVALID_ARG, "get_option: NULL argument");
    if (c->multi() && strcmp(name, "level_split_min") && strcmp(name, "level_exchange")) {
        const int32_t rc = tfhe_get_option(c->kids[0], name, value);      // set_option gives every device the same value
        if (rc) c->err = c->kids[0]->err;
        return rc;
    }
    const struct { const char *n; int64_t v; } table[] = {
        {"br_small", c->br_small}, {"br_tiny", c->br_tiny}, {"br_rt_l", c->br_rt_l}, {"timing_events", c->timing_events},
        {"br_split", c->br_split}, {"br_general", c->br_general}, {"br_anyn", c->br_anyn}, {"anyn_spec", c->anyn_spec},
        {"level_split_min", c->level_split_min}, {"level_exchange", c->level_exchange}, {"br_prio_pct", c->br_prio_pct}, {"ks_slices", c->ks_slices_large},
        {"measure_margin", c->measure_margin ? 1 : 0}, {"pipeline_min", c->pipeline_min}, {"w2_rw", c->w2_rw}, {"k2_rw", c->k2_rw}, {"k2_w3", c->k2_w3}, {"n512_rw", c->n512_rw}, {"n512_w2", c->n512_w2},
        {"v3_rw", c->v3_rw}, {"mk_general", c->mk_force_general ? 1 : 0}, {"n2048_rw", c->n2048_rw}, {"mkg_acc", c->mkg_acc},
        {"mkg_variant", c->mkg_variant}, {"mkg_rw", c->mkg_rw}, {"mk_rw", c->mk_rw}, {"ks_variant", c->ks_variant},
    };
    for (const auto &e : table)
        if (!strcmp(name, e.n)) { *value = e.v; return TFHE_OK; }
    return c->set_err(TFHE_ERR_INVALID_ARG, "get_option: unknown option '%s'", name);
}

int32_t tfhe_set_option(tfhe_ctx *c, const char *name, int64_t value)
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    if (!name || !*name) return TFHE_OK;
    if (c->multi()) {
        if (!strcmp(name, "level_split_min")) { c->level_split_min = value; return TFHE_OK; }       // the options that belong to the fan-out context itself
        if (!strcmp(name, "level_exchange")) {
            if (value < 0 || value > 2) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: level_exchange must be 0 (by peer access), 1 (device-to-device) or 2 (host staging)");
            c->level_exchange = (int)value;
            return TFHE_OK;
        }
        for (tfhe_ctx *k : c->kids) {
            const int32_t rc = tfhe_set_option(k, name, value);
            if (rc) { c->err = k->err; return rc; }
        }
        return TFHE_OK;
    }
    if (!strcmp(name, "br_small")) { c->br_small = value; return TFHE_OK; }
    if (!strcmp(name, "br_tiny")) { c->br_tiny = value; return TFHE_OK; }
    if (!strcmp(name, "br_rt_l")) { c->br_rt_l = value != 0; return TFHE_OK; }
    if (!strcmp(name, "timing_events")) { c->timing_events = value != 0; return TFHE_OK; }
    if (!strcmp(name, "br_split")) { c->br_split = value != 0; return TFHE_OK; }
    if (!strcmp(name, "br_general")) { c->br_general = value != 0; return TFHE_OK; }
    if (!strcmp(name, "br_anyn")) {
        // the any-N kernels read the bootstrapping key in their own spectrum order: choose before loading it
        if ((c->have_bk || c->have_mk_bk) && (value != 0) != (c->br_anyn != 0))
            return c->set_err(TFHE_ERR_STATE, "set_option: br_anyn must be chosen before the bootstrapping key is loaded (reload the key after changing it)");
        c->br_anyn = value != 0;
        return TFHE_OK;
    }
    if (!strcmp(name, "anyn_spec")) {
        if (value < -1 || value > 1) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: anyn_spec must be -1 (LDS when it fits), 0 (LDS) or 1 (global memory)");
        c->anyn_spec = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "level_split_min") || !strcmp(name, "level_exchange")) return TFHE_OK;      // (meaningful on a multi-device context only)
    if (!strcmp(name, "br_prio_pct")) {
        if (value < 0 || value > 100) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: br_prio_pct must be 0..100");
        c->br_prio_pct = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "ks_slices")) {
        if (value != 1 && value != 2 && value != 4) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: ks_slices must be 1, 2 or 4");
        c->ks_slices_large = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "measure_margin")) { c->measure_margin = value != 0; return TFHE_OK; }
    if (!strcmp(name, "pipeline_min")) { c->pipeline_min = value; return TFHE_OK; }
    if (!strcmp(name, "w2_rw")) {
        if (value < 0 || value > 2) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: w2_rw must be 0 (by batch size), 1 or 2");
        c->w2_rw = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "n512_w2")) {
        if (value < -1 || value > 1) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: n512_w2 must be -1 (by batch size), 0 (never) or 1 (always)");
        c->n512_w2 = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "n512_rw")) {
        if (value != 0 && value != 1 && value != 4) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: n512_rw must be 0 (by batch size), 1 or 4");
        c->n512_rw = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "k2_w3")) {
        if (value < -1 || value > 1) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: k2_w3 must be -1 (by batch size), 0 (never) or 1 (always)");
        c->k2_w3 = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "k2_w3_cost_us")) {      // measurement aid: the round cost the partition prices the three-wave tail at (microseconds)
        if (value < 0) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: k2_w3_cost_us must be >= 0");
        kK2W3RoundCost[1] = (double)value / 1000.0; kK2W3RoundCost[2] = (double)value / 1000.0 + 0.1;
        return TFHE_OK;
    }
    if (!strcmp(name, "k2_rw")) {
        if (value != 0 && value != 1 && value != 7) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: k2_rw must be 0 (by batch size), 1 or 7");
        c->k2_rw = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "v3_rw")) {
        if (value != 0 && value != 1 && value != 4) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: v3_rw must be 0 (by batch size), 1 or 4");
        c->v3_rw = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "mk_general")) { c->mk_force_general = value != 0; return TFHE_OK; }
    if (!strcmp(name, "n2048_rw")) {
        if (value != 0 && value != 1 && value != 2) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: n2048_rw must be 0 (by batch size), 1 or 2");
        c->n2048_rw = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "mkg_acc")) {
        if (value < -1 || value > 1) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: mkg_acc must be -1 (by party count), 0 (LDS) or 1 (global memory)");
        c->mkg_acc = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "mkg_variant")) {
        if (value != 0 && value != 1) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: mkg_variant must be 0 (two-wave kernel for the shipped 4- / 8-party sets) or 1 (any-party kernel)");
        c->mkg_variant = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "mkg_rw")) {
        if (value != 0 && value != 1 && value != 2 && value != 4) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: mkg_rw must be 0 (by batch size), 1, 2 or 4 (4: the two-wave 4- / 8-party kernel only)");
        c->mkg_rw = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "mk_rw")) {
        if (value != 0 && value != 1 && value != 2) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: mk_rw must be 0 (by batch size), 1 or 2");
        c->mk_rw = (int)value;
        return TFHE_OK;
    }
    if (!strcmp(name, "ks_variant")) {
        if (value != 1 && value != 3 && value != 4) return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: ks_variant must be 1, 3 or 4");
        // only the selected family's key layout is kept on the device: choose before loading the keyswitch key
        if ((c->have_ks || c->have_mk_ks) && (int)value != c->ks_variant)
            return c->set_err(TFHE_ERR_STATE, "set_option: ks_variant must be chosen before the keyswitch key is loaded (reload the key after changing it)");
        c->ks_variant = (int)value;
        return TFHE_OK;
    }
    return c->set_err(TFHE_ERR_INVALID_ARG, "set_option: unknown option '%s'", name);
}

}  // extern "C"
