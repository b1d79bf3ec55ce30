// engine_keys.hip — keys: bootstrapping / keyswitch key loaders (single- and multi-key), key generation and RGSW.Expand on the device
#define TFHE_EMIT_KEYPREP_KERNELS
#include "engine.hpp"
#include "kernels_keygen.hpp"

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
void quiesce(tfhe_ctx *c)
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

int32_t tfhe_load_bootstrap_key_i32(tfhe_ctx *c, const int32_t *bk) try
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    return load_bk_common(c, bk, bk_poly_count(c->P) * (size_t)c->P.N * sizeof(int32_t), false);
}
ABI_CATCH(c, "tfhe_load_bootstrap_key_i32")

int32_t tfhe_load_bootstrap_key_c128(tfhe_ctx *c, const double *bk_spectra) try
{
    ENTER_CTX(c);
    if (!c) return TFHE_ERR_INVALID_ARG;
    return load_bk_common(c, bk_spectra, bk_poly_count(c->P) * (size_t)(c->P.N / 2) * sizeof(cplx), true);
}
ABI_CATCH(c, "tfhe_load_bootstrap_key_c128")

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

int32_t tfhe_load_keyswitch_key(tfhe_ctx *c, const int32_t *ks) try
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
ABI_CATCH(c, "tfhe_load_keyswitch_key")

// Generates the cloud key on the device (kernels_keygen.hpp) and loads it: the analogue of CloudKey(rng, secret_key)
// (api.jl:111-127) with the secret material supplied by the caller.  Optionally copies the canonical Int32 arrays back.
int32_t tfhe_keygen_cloud_key(tfhe_ctx *c, const int32_t *lwe_key, const int32_t *tlwe_key, double bs_noise_stddev,
                              double ks_noise_stddev, const uint32_t *seed, int32_t *bk_out, int32_t *ks_out) try
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
        alloc_checkpoint();
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
    // (also when body() ends in an exception — a staging vector that could not be allocated: the secret material is zeroed and freed)
    auto guard = on_exit([&] { (void)hipSetDevice(g->device); cleanup(); });
    return body();
}
ABI_CATCH(c, "tfhe_keygen_cloud_key")

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

int32_t tfhe_mk_load_bootstrap_key_i32(tfhe_ctx *c, const int32_t *bk, int32_t parties) try
{
    return mk_load_bk_common(c, bk, parties, false);
}
ABI_CATCH(c, "tfhe_mk_load_bootstrap_key_i32")

int32_t tfhe_mk_load_bootstrap_key_c128(tfhe_ctx *c, const double *bk_spectra, int32_t parties) try
{
    return mk_load_bk_common(c, bk_spectra, parties, true);
}
ABI_CATCH(c, "tfhe_mk_load_bootstrap_key_c128")

// RGSW.Expand on the device (mk_internals.jl:304-345, MKBootstrapKey :442-461): the parties' uni-encryptions and public
// keys in, the expanded transformed bootstrapping key resident on the device out.
int32_t tfhe_mk_expand_load_bootstrap_key(tfhe_ctx *c, int32_t parties, const int32_t *pub_b, const int32_t *c0, const int32_t *c1,
                                          const int32_t *d0, const int32_t *d1, const int32_t *f0, const int32_t *f1, int32_t *expanded_out) try
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
        alloc_checkpoint();
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
    int32_t rc;
    {
        auto guard = on_exit(cleanup);      // (also when body() ends in an exception: h_dec is gigabytes at hundreds of parties)
        rc = body();
    }
    if (rc) return rc;
    c->mk_parties = parties;
    c->have_mk_bk = true;
    return TFHE_OK;
}
ABI_CATCH(c, "tfhe_mk_expand_load_bootstrap_key")

int32_t tfhe_mk_load_keyswitch_key(tfhe_ctx *c, const int32_t *ks, int32_t parties) try
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
    c->mk_ks_parties = parties;
    c->have_mk_ks = true;
    return TFHE_OK;
}
ABI_CATCH(c, "tfhe_mk_load_keyswitch_key")

