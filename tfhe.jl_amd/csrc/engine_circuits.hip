// engine_circuits.hip — circuits: the device-resident wire table, one level of gates per call, rows travelling between the devices of a context
#include "engine.hpp"
#include <unordered_set>

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

// ---- levelised circuit execution on a device-resident wire table -------------------------
// On a multi-device context every device holds a replica of the wire table; the context tracks which replicas hold each
// wire's current value (wire_valid / wire_owner) and pull_wires brings a device up to date, device to device, for exactly the
// rows it is about to read.
int32_t tfhe_wires_alloc(tfhe_ctx *c, int64_t num_wires) try
{
    ENTER_CTX(c);
    if (c->multi()) {      // one replica of the table per device
        const int32_t rc = fan_out(c, all_kids(c), [&](int k) { return tfhe_wires_alloc(c->kids[(size_t)k], num_wires); });
        if (rc == TFHE_OK) {
            c->num_wires = num_wires;
            const size_t nw = (size_t)std::max<int64_t>(num_wires, 0);
            alloc_checkpoint();
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
ABI_CATCH(c, "tfhe_wires_alloc")

static int32_t wires_range_ok(tfhe_ctx *c, const char *who, int64_t first, int64_t count, const void *host)
{
    if (!c->multi() && !c->d_wires) return c->set_err(TFHE_ERR_STATE, "%s: no wire table allocated", who);
    if (c->multi() && c->num_wires <= 0) return c->set_err(TFHE_ERR_STATE, "%s: no wire table allocated", who);
    if (first < 0 || count < 0 || first + count > c->num_wires || (count > 0 && !host))
        return c->set_err(TFHE_ERR_INVALID_ARG, "%s: wire range [%lld, %lld) outside the table of %lld wires or NULL buffer", who,
                          (long long)first, (long long)(first + count), (long long)c->num_wires);
    return TFHE_OK;
}

int32_t tfhe_wires_upload(tfhe_ctx *c, int64_t first, int64_t count, const int32_t *host) try
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
ABI_CATCH(c, "tfhe_wires_upload")

int32_t tfhe_wires_download(tfhe_ctx *c, int64_t first, int64_t count, int32_t *host) try
{
    ENTER_CTX(c);
    if (c->multi()) {      // the first device is brought up to date for the range, then read
        int32_t rc = wires_range_ok(c, "wires_download", first, count, host);
        if (rc || count == 0) return rc;
        alloc_checkpoint();
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
ABI_CATCH(c, "tfhe_wires_download")


int32_t tfhe_wires_gather(tfhe_ctx *c, const int32_t *wires, int64_t count, int32_t *host) try
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
ABI_CATCH(c, "tfhe_wires_gather")

// every index in range; no wire both written and read inside one level (the level's gates are independent).
// O(B) work whatever the size of the wire table: only the level's own output wires are hashed.
static int32_t validate_level(tfhe_ctx *c, int64_t num_wires, const uint8_t *opcodes, const int32_t *a, const int32_t *b, const int32_t *cc,
                              const int32_t *out, int64_t B)
{
    alloc_checkpoint();
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
int32_t pull_wires(tfhe_ctx *c, int dst, const int32_t *wires, int64_t count)
{
    const int nk = (int)c->kids.size();
    alloc_checkpoint();
    std::vector<std::vector<int32_t>> from((size_t)nk);
    std::vector<uint8_t> &valid = c->wire_valid[(size_t)dst];
    // (two passes: the rows are marked valid only once every list has been built — a push_back that throws half-way must not
    //  leave rows marked that nobody is going to fetch)
    for (int64_t i = 0; i < count; i++) {
        const int32_t w = wires[i];
        if (w < 0 || valid[(size_t)w]) continue;
        from[(size_t)c->wire_owner[(size_t)w]].push_back(w);
    }
    for (int src = 0; src < nk; src++) {
        std::vector<int32_t> &rows = from[(size_t)src];
        size_t keep = 0;
        for (int32_t w : rows)                              // drops duplicates in the list
            if (!valid[(size_t)w]) { valid[(size_t)w] = 1; rows[keep++] = w; }
        rows.resize(keep);
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
            if (!px) px = new (std::nothrow) PairXfer();
            if (!px) return c->set_err(TFHE_ERR_NOMEM, "gates_level: out of host memory (transfer ring of devices %d -> %d)", ks->device, kd->device);
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
    alloc_checkpoint();
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
                         const int32_t *out, int64_t B) try
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
ABI_CATCH(c, "tfhe_gates_level")
