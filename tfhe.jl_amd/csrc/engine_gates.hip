// engine_gates.hip — gates: the batch entry points (device pointers, host buffers, streaming), bootstrap and keyswitch batches
#include "engine.hpp"
#include <chrono>
#include "kernels_gates.hpp"

// Workspaces (bara, ext, abar, map, diag) are per context: work a previous call enqueued (possibly on another stream)
// must have finished before a call on stream `s` reuses them.  The previous call recorded done_ev at its end; making
// `s` wait for that event orders the two without blocking the host and without keeping the caller's stream handle.
int32_t enter_stream(tfhe_ctx *c, hipStream_t s)
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
int32_t leave_stream(tfhe_ctx *c, hipStream_t s)
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
void next_timing_slot(tfhe_ctx *c)
{
    c->ev = c->evring[c->timed_calls % tfhe_ctx::kTimingSlots];
    c->timing_valid = false;
    c->last_call_two_streams = false;
}
void commit_timing_slot(tfhe_ctx *c)
{
    c->timed_calls++;
    c->timing_valid = true;
}

int32_t ensure_host_map(tfhe_ctx *c, size_t bytes)
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

// Common body of tfhe_gates_batch_dev (operands = rows g of three arrays, ia = ib = ic = io = NULL) and
// tfhe_gates_level (operands = rows ia[g], ib[g], ic[g] of one wire table, result row io[g]; operands an opcode
// does not read are replaced by row 0).
int32_t run_gates(tfhe_ctx *c, const char *who, const uint8_t *opcodes, int64_t B, const int32_t *d_in0,
                  const int32_t *d_in1, const int32_t *d_in2, int32_t *d_out, const int32_t *ia, const int32_t *ib,
                  const int32_t *ic, const int32_t *io, hipStream_t s)
{
#ifdef TFHE_EXPERIMENT
    // development builds (experiment.hpp), TFHE_DEBUG_HOSTTIME=1: host microseconds per section of this function, printed per call — what
    // showed that a circuit level's host time was the wait for the previous level's staging copy, not anything in here (round 5)
    static const bool dbg_host = getenv("TFHE_DEBUG_HOSTTIME") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    double t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    auto MARK = [&](int k) { if (dbg_host) { const auto t = std::chrono::steady_clock::now(); t_acc[k] += std::chrono::duration<double, std::micro>(t - t_prev).count(); t_prev = t; } };
#else
    auto MARK = [](int) {};
#endif
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
        rc = launch_prologue(c, R, d_in0, d_in1, d_in2, d_ra, d_rb, d_kind, n, s);
        if (rc) return rc;
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
#ifdef TFHE_EXPERIMENT
    if (dbg_host) fprintf(stderr, "run_gates host us: classify+enter %.0f | stage %.0f | fill %.0f | map H2D %.0f | map event %.0f | prologue %.0f | blind rotate %.0f | keyswitch+trivial %.0f\n",
                          t_acc[0], t_acc[1], t_acc[2], t_acc[3], t_acc[4], t_acc[5], t_acc[6], t_acc[7]);
#endif
    return leave_stream(c, s);
}

// gate prologue (gates.jl) + modulus switch (bootstrap.jl:74-75) of R rotations into the context's bara workspace; `words` = mask
// words per sample (n; multi-key: P n)
int32_t launch_prologue(tfhe_ctx *c, size_t R, const int32_t *d_in0, const int32_t *d_in1, const int32_t *d_in2, const int32_t *d_ra, const int32_t *d_rb,
                        const uint8_t *d_kind, int words, hipStream_t s)
{
    hipLaunchKernelGGL(prologue_kernel, dim3((unsigned)R), dim3(256), 0, s, d_in0, d_in1, d_in2, d_ra, d_rb, d_kind, (int32_t *)c->bara.p, words, ilog2i(2 * c->P.N));
    HIP_TRY(c, hipGetLastError());
    return TFHE_OK;
}

int32_t tfhe_gates_batch_dev(tfhe_ctx *c, const uint8_t *opcodes, const int32_t *d_in0, const int32_t *d_in1,
                             const int32_t *d_in2, int32_t *d_out, int64_t B, void *stream) try
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
ABI_CATCH(c, "tfhe_gates_batch_dev")

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
        c->twin_stream.store(t->stream, std::memory_order_release);
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
    alloc_checkpoint();
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
                         const int32_t *in2, int32_t *out, int64_t B) try
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
                second = std::thread([&] {
                    try { rc1 = gates_batch_one_stream(c->twin, opcodes + h, off(in0, h), off(in1, h), off(in2, h), out + (size_t)h * n1, B - h, need); }
                    catch (...) { rc1 = abi_caught(c->twin, "gates_batch (second half)"); }
                });
            } catch (const std::exception &) {
                return gates_batch_one_stream(c, opcodes, in0, in1, in2, out, B, need);
            }
            int32_t rc0;
            {
                auto joined = on_exit([&] { second.join(); });      // (also when the first half ends in an exception: a joinable std::thread must not be destroyed)
                rc0 = gates_batch_one_stream(c, opcodes, in0, in1, in2, out, h, need);
            }
            if (rc1) return c->set_err(rc1, "gates_batch (second half): %s", c->twin->err.c_str());
            if (rc0) return rc0;
            c->last_rotations += c->twin->last_rotations;
            c->last_call_two_streams = true;
            return TFHE_OK;
        }
    }
    return gates_batch_one_stream(c, opcodes, in0, in1, in2, out, B, need);
}
ABI_CATCH(c, "tfhe_gates_batch")

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
                                const int32_t *in2, int32_t *out, int64_t B, int32_t *ticket) try
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
        alloc_checkpoint();
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
        c->submit_thread.store(std::this_thread::get_id(), std::memory_order_release);
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
    c->submit_thread.store(std::this_thread::get_id(), std::memory_order_release);
    *ticket = slot;
    return TFHE_OK;
}
ABI_CATCH(c, "tfhe_gates_batch_submit")

// Callable from any thread, also while another thread is inside a call on the context (ABI v7).  From the thread that submitted:
// the wait for the ticket's stream, and the slot is released.  From ANY OTHER thread — a finalizer, a task that migrated — it
// never takes the context (so it cannot make the submitting thread's next call fail with TFHE_ERR_STATE either): it gets the
// unguarded wait of tfhe_ctx_synchronize — everything queued so far, a superset of the ticket's batch — and the slot stays
// marked until the submitting thread's next submit or wait finds the stream idle.  (Until v6 a wait that overlapped another
// thread's call returned TFHE_ERR_STATE at once, and a caller that then freed the batch's page-locked buffers — a finalizer —
// freed them under a live DMA.)
int32_t tfhe_gates_batch_wait(tfhe_ctx *c, int32_t ticket) try
{
    if (!c) return TFHE_ERR_INVALID_ARG;
    alloc_checkpoint();
    const std::thread::id submitter = c->submit_thread.load(std::memory_order_acquire);
    if (submitter != std::thread::id{} && submitter != std::this_thread::get_id() && c->owner.load(std::memory_order_acquire) != std::this_thread::get_id()) {
        if (ticket != 0 && ticket != 1 && ticket != 2) return TFHE_ERR_INVALID_ARG;
        return ticket == 2 ? TFHE_OK : tfhe_ctx_synchronize(c);
    }
    CallGuard call_guard_(c);
    if (!call_guard_.ok) {
        if (g_rejected_ctx == c) g_rejected_ctx = nullptr;      // (not an error here)
        if (ticket != 0 && ticket != 1 && ticket != 2) return TFHE_ERR_INVALID_ARG;
        return ticket == 2 ? TFHE_OK : tfhe_ctx_synchronize(c);
    }
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
ABI_CATCH(c, "tfhe_gates_batch_wait")

int32_t tfhe_bootstrap_batch(tfhe_ctx *c, int32_t mu, const int32_t *in, int32_t *out, int64_t B, int32_t with_keyswitch) try
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
ABI_CATCH(c, "tfhe_bootstrap_batch")

int32_t tfhe_keyswitch_batch(tfhe_ctx *c, const int32_t *in, int32_t *out, int64_t B) try
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
ABI_CATCH(c, "tfhe_keyswitch_batch")
