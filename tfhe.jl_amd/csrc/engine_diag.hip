// engine_diag.hip — diagnostics: timing of the last calls, rounding margin, in-kernel clock, kernel names, options
#include "engine.hpp"

int32_t tfhe_last_timing_ms(tfhe_ctx *c, int32_t which, float *ms) try
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
ABI_CATCH(c, "tfhe_last_timing_ms")

int32_t tfhe_timing_history_ms(tfhe_ctx *c, int32_t which, float *ms, int32_t max_calls, int32_t *n_out) try
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
ABI_CATCH(c, "tfhe_timing_history_ms")

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
#ifdef TFHE_EXPERIMENT
        if (getenv("TFHE_DEBUG_LIFETIMES")) {       // development builds (experiment.hpp): wave lifetimes from the 100 MHz s_memrealtime counter
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
#endif
        std::nth_element(f.begin(), f.begin() + (long)(f.size() / 2), f.end());
        *mhz = f[f.size() / 2];
    }
    return TFHE_OK;
}

int32_t tfhe_last_rounding_margin(tfhe_ctx *c, double *worst) try
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
ABI_CATCH(c, "tfhe_last_rounding_margin")

int32_t tfhe_last_kernel_clock_mhz(tfhe_ctx *c, double *mhz) try
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
ABI_CATCH(c, "tfhe_last_kernel_clock_mhz")

#ifdef TFHE_STAMP
// development builds only (make stamp): per-phase shader-clock ticks of workgroup 0's waves, [4 waves][16]
extern "C" int32_t tfhe_debug_phases(tfhe_ctx *c, unsigned long long *out64)
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
int32_t tfhe_get_option(tfhe_ctx *c, const char *name, int64_t *value) try
{
    if (!c && name && value && !strcmp(name, "debug_fail_alloc_after")) { *value = g_fail_alloc_countdown.load(); return TFHE_OK; }
    ENTER_CTX(c);
    if (!name || !value) return c->set_err(TFHE_ERR_INVALID_ARG, "get_option: NULL argument");
    // what tfhe_ctx_create decided about this parameter set's Float64 exactness (engine_context.hip: exactness_class)
    if (!strcmp(name, "exact_domain")) { *value = c->exact_domain; return TFHE_OK; }
    if (!strcmp(name, "exact_bound_log2_x1000")) { *value = (int64_t)std::llround(c->exact_bound_log2 * 1000.0); return TFHE_OK; }
    if (!strcmp(name, "exact_margin_x1e6")) { *value = (int64_t)std::llround(c->exact_margin * 1e6); return TFHE_OK; }
    if (!strcmp(name, "debug_fail_alloc_after")) { *value = g_fail_alloc_countdown.load(); return TFHE_OK; }
    if (c->multi() && !strcmp(name, "peer_pairs")) {      // read-only: ordered pairs of DIFFERENT device contexts of this context that copy device to device
        int64_t n = 0;
        const size_t nk = c->kids.size();
        for (size_t a = 0; a < nk; a++)
            for (size_t b = 0; b < nk; b++) n += (a != b && c->peer_ok[a * nk + b]) ? 1 : 0;
        *value = n;
        return TFHE_OK;
    }
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
ABI_CATCH(c, "tfhe_get_option")

int32_t tfhe_set_option(tfhe_ctx *c, const char *name, int64_t value) try
{
    // process-wide, settable without a context (tests/test_abi_nomem.py): the n-th allocation checkpoint from now on throws
    // std::bad_alloc (engine.hpp: alloc_checkpoint); 0 switches it off
    if (name && !strcmp(name, "debug_fail_alloc_after")) {
        if (value < 0) return TFHE_ERR_INVALID_ARG;
        g_fail_alloc_countdown.store(value);
        return TFHE_OK;
    }
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
ABI_CATCH(c, "tfhe_set_option")
