// engine.hpp — what the host-side translation units of libtfhe_mi355x.so share: the context, the status / guard macros and the
// declarations of the functions one section calls in another.  C ABI: include/tfhe_mi355x.h.
//
//   engine_context.hip    create / destroy, tables, sharding rule, host memory, last error            ("context")
//   engine_keys.hip       bootstrapping / keyswitch key loaders, device keygen, RGSW.Expand            ("keys")
//   engine_dispatch.hip   which blind-rotate / keyswitch kernel a batch takes, and its launch          ("dispatch")
//   engine_gates.hip      tfhe_gates_batch*, tfhe_bootstrap_batch, tfhe_keyswitch_batch, streaming     ("gates")
//   engine_circuits.hip   wire table, tfhe_gates_level, rows between the devices of a context         ("circuits")
//   engine_multikey.hip   tfhe_mk_gate_nand_batch and its kernels                                     ("multi-key")
//   engine_diag.hip       timing, rounding margin, in-kernel clock, options                           ("diagnostics")
//   mk_g2_inst.hip        one instantiation of the many-party two-wave kernel per translation unit
//
// Every kernel is compiled in exactly one translation unit: the kernel headers define their non-template __global__ functions
// only under the TFHE_EMIT_* macro of the unit that launches them (templates are instantiated where they are launched).
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
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tfhe_mi355x.h"
#include "experiment.hpp"
#include "br_core.hpp"

using namespace tfhe;

#include "kernels_blind_rotate.hpp"
#include "kernels_anyn.hpp"
#include "kernels_n512.hpp"
#include "kernels_keyswitch.hpp"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
// Error text without allocation: an error path (and the handler that turns std::bad_alloc into TFHE_ERR_NOMEM) must not allocate.
struct ErrBuf {
    char s[512] = {0};
    const char *c_str() const { return s; }
    void set(const char *msg) { snprintf(s, sizeof s, "%s", msg ? msg : ""); }
    ErrBuf &operator=(const char *msg) { set(msg); return *this; }
    void format(const char *fmt, ...) __attribute__((format(printf, 2, 3)))
    {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(s, sizeof s, fmt, ap);
        va_end(ap);
    }
};
extern thread_local ErrBuf g_create_error;      // tfhe_last_error(NULL): the last failure of a call that has no context (this thread's)

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
    ErrBuf err;
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
    int mk_parties = 0;            // parties of the loaded multi-key bootstrapping key
    int mk_ks_parties = 0;         // ... and of the loaded multi-key keyswitch key (tfhe_mk_gate_nand_batch needs them equal)
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
    // Float64 exactness domain of this parameter set, decided at creation (exactness_class, engine_context.hip;
    // tfhe_get_option "exact_domain", "exact_bound_log2_x1000", "exact_margin_x1e6")
    int exact_domain = 2;          // 2 = guaranteed exact for every key, 1 = exact for real (uniform) keys, 0 = outside
    double exact_bound_log2 = 0;   // log2 of the a-priori worst-case |pre-rounding value| (any Int32 key words)
    double exact_margin = 0;       // predicted rounding margin with a real key (distance from an integer; 0.5 flips a rounding)
    // Host-buffer batches of at least `pipeline_min` gates are cut in two rotation-balanced halves that run on two streams of
    // this device, so that the second half's upload and the first half's download cross PCIe while the other half computes.
    // The second stream, its workspaces and its events belong to a twin context that BORROWS this context's keys.
    tfhe_ctx *twin = nullptr;
    std::atomic<hipStream_t> twin_stream{nullptr};      // the twin's stream, published once by ensure_twin: tfhe_ctx_synchronize reads it from any thread
    bool borrows_keys = false;
    bool slot_busy[2] = {false, false};   // tfhe_gates_batch_submit: a batch is in flight on the own (0) / the twin's (1) stream
    std::atomic<std::thread::id> submit_thread{};      // the thread of the last tfhe_gates_batch_submit: ITS waits release the slot, any other thread's wait only waits
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
        ErrBuf next;                  // (the arguments may point into `err` itself: a kid's message wrapped into its parent's)
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(next.s, sizeof next.s, fmt, ap);
        va_end(ap);
        err = next;
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
extern thread_local const tfhe_ctx *g_rejected_ctx;
extern const char kBusyMessage[];
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
    alloc_checkpoint();                                                                            \
    CallGuard call_guard_(ctx);                                                                    \
    if (!call_guard_.ok) return TFHE_ERR_STATE

// ---- no exception crosses the C ABI ------------------------------------------------------------------------------------
// Every extern "C" entry point is a function-try-block closed by ABI_CATCH: std::bad_alloc (a std::vector that could not grow, a
// std::thread that could not start, ...) becomes TFHE_ERR_NOMEM, anything else TFHE_ERR_STATE with what(); the message goes to a
// thread-local buffer that tfhe_last_error(ctx) returns to THIS thread, and into the context's own buffer if the context can be
// taken again (the CallGuard has been released by the unwinding: if another thread has entered since, the buffer is theirs) —
// that copy is what the parent of a library-owned worker thread, or of a device context, reads.  The context stays usable: workspaces are only
// ever grown (DevBuf), a half-queued call is ordered by the stream, and every loader clears its have_* flag before it starts.
int32_t abi_caught(tfhe_ctx *c, const char *who) noexcept;
#define ABI_CATCH(ctx, who) catch (...) { return abi_caught(ctx, who); }
// Allocation-failure injection for the tests of that promise: tfhe_set_option(ctx or NULL, "debug_fail_alloc_after", n) makes the
// n-th allocation checkpoint from then on, process-wide, throw std::bad_alloc (0 = off, the default).  Checkpoints sit at the
// start of every entry point and in front of the host allocations that grow with the batch (index maps, shard tables, level
// validation, wire bookkeeping).
extern std::atomic<int64_t> g_fail_alloc_countdown;
inline void alloc_checkpoint()
{
    if (g_fail_alloc_countdown.load(std::memory_order_relaxed) > 0 && g_fail_alloc_countdown.fetch_sub(1, std::memory_order_relaxed) == 1) throw std::bad_alloc();
}

// ---- hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel), never lowered ---------------------------
// A driver call on the host side of every launch of a kernel that needs more than 64 KB of LDS — on the path where latency is
// the metric (a single gate, a circuit level).  The attribute belongs to (device, function), is shared by every context of the
// process and only ever needs to grow (the any-N kernels' LDS depends on the parameter set), so the largest value set so far
// is remembered process-wide.
int32_t ensure_dyn_lds(tfhe_ctx *c, const void *fn, size_t bytes, const char *what);
#define LDS_TRY(ctx, bytes, ...)                                                                   \
    do {                                                                                           \
        const int32_t rl_ = ensure_dyn_lds(ctx, (const void *)(__VA_ARGS__), (size_t)(bytes), #__VA_ARGS__); \
        if (rl_) return rl_;                                                                       \
    } while (0)

constexpr size_t kH2TableOffset = kTableElems + 1024;      // tw1h | tw2q | tw3q of blind_rotate_kernel_h2
constexpr size_t kN512TableOffset = kH2TableOffset + kH2TableElems;      // tw1 of blind_rotate_kernel_n512

inline int ilog2i(int x) { int r = 0; while ((1 << r) < x) r++; return r; }
inline bool op_has_a(int op) { return !(op == TFHE_GATE_CONST0 || op == TFHE_GATE_CONST1); }
inline bool op_has_b(int op) { return op_has_a(op) && !(op == TFHE_GATE_NOT || op == TFHE_GATE_COPY); }
inline size_t bk_poly_count(const tfhe_params &p) { return (size_t)p.n * p.bs_l * (p.k + 1) * (p.k + 1); }

// runs f when the scope ends, by return or by exception
template <typename F>
struct ScopeExit {
    F f;
    ~ScopeExit() { f(); }
};
template <typename F>
ScopeExit<F> on_exit(F f) { return ScopeExit<F>{f}; }

// ---- fan-out helpers --------------------------------------------------------------------------------
// Runs fn(kid index) for every kid in `which` concurrently (one library-owned thread per extra kid; the calling
// thread takes the first) and returns the first failing status, copying that kid's message.
template <typename F>
static int32_t fan_out(tfhe_ctx *c, const std::vector<int> &which, F &&fn)
{
    // Everything that can throw is allocated BEFORE the first thread exists, and every shard — on a worker thread or on the
    // calling one — turns an exception into its status: a std::thread that is still joinable when an exception unwinds this frame
    // would end the process (std::terminate), as would an exception that left a thread's function.
    std::vector<int32_t> rcs(which.size(), TFHE_OK);
    std::vector<std::thread> th;
    std::vector<size_t> inline_ones;          // kids whose thread could not be started (thread / process limit): run here, in turn
    th.reserve(which.size());
    inline_ones.reserve(which.size());
    auto shard = [&](size_t i) noexcept {
        try { rcs[i] = fn(which[i]); } catch (...) { rcs[i] = abi_caught(c->kids[(size_t)which[i]], "fan-out shard"); }
    };
    for (size_t i = 1; i < which.size(); i++) {
        try {
            th.emplace_back([&shard, i] { shard(i); });
        } catch (...) {                       // the thread could not be started (std::system_error): this shard runs on the caller's thread
            inline_ones.push_back(i);
        }
    }
    if (!which.empty()) shard(0);
    for (size_t i : inline_ones) shard(i);
    for (auto &t : th) t.join();
    std::fill(c->kid_ran.begin(), c->kid_ran.end(), 0);
    for (int k : which) c->kid_ran[(size_t)k] = 1;
    for (size_t i = 0; i < which.size(); i++)
        if (rcs[i]) return c->set_err(rcs[i], "device %d (kid %d): %s", c->kids[(size_t)which[i]]->device, which[i], c->kids[(size_t)which[i]]->err.c_str());
    return TFHE_OK;
}
inline std::vector<int> all_kids(const tfhe_ctx *c)
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

// ---- what one section calls in another ---------------------------------------------------------------------------------
// engine_context.hip
void shard_bounds_by_rotations(const uint8_t *opcodes, int64_t B, int shards, int64_t *bounds);
// engine_keys.hip
void quiesce(tfhe_ctx *c);
// engine_dispatch.hip
struct DiagArgs;
int32_t prepare_diag(tfhe_ctx *c, size_t R, hipStream_t s, DiagArgs &d);
void name_kernel(tfhe_ctx *c, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int32_t launch_blind_rotate(tfhe_ctx *c, size_t R, int32_t mu, hipStream_t s);
int32_t launch_keyswitch(tfhe_ctx *c, size_t G, const int32_t *e0, const int32_t *e1, const int32_t *dst, const int32_t *ext, int32_t *out, hipStream_t s);
int32_t launch_mk_keyswitch(tfhe_ctx *c, size_t B, const int32_t *d_gate, int32_t *out, hipStream_t s);
// engine_gates.hip
int32_t enter_stream(tfhe_ctx *c, hipStream_t s);
int32_t leave_stream(tfhe_ctx *c, hipStream_t s);
void next_timing_slot(tfhe_ctx *c);
void commit_timing_slot(tfhe_ctx *c);
int32_t ensure_host_map(tfhe_ctx *c, size_t bytes);
int32_t launch_prologue(tfhe_ctx *c, size_t R, const int32_t *d_in0, const int32_t *d_in1, const int32_t *d_in2, const int32_t *d_ra, const int32_t *d_rb,
                        const uint8_t *d_kind, int words, hipStream_t s);
int32_t run_gates(tfhe_ctx *c, const char *who, const uint8_t *opcodes, int64_t B, const int32_t *d_in0, const int32_t *d_in1, const int32_t *d_in2,
                  int32_t *d_out, const int32_t *ia, const int32_t *ib, const int32_t *ic, const int32_t *io, hipStream_t s);
// engine_circuits.hip
int32_t pull_wires(tfhe_ctx *c, int dst, const int32_t *wires, int64_t count);
