/*
 * tfhe_mi355x.h — C ABI of the MI355X-native TFHE gate-bootstrapping engine (libtfhe_mi355x.so).
 *
 * The reference (nucypher/TFHE.jl) has no FFI for this path: it is plain Julia.  This header is the
 * drop-in boundary a `ccall` shim binds instead of the Julia functions cited per entry point
 * (paths relative to the reference checkout).  Plain pointers and sizes only; no C++ types, no
 * exceptions cross the boundary: every entry point catches what its C++ body throws (a std::vector or std::thread that could
 * not be had: TFHE_ERR_NOMEM; anything else: TFHE_ERR_STATE with the exception's text) and the context stays usable — under a
 * Julia `ccall` or Python `ctypes` an escaping exception would end the process.  Every function returns 0 on success or a
 * TFHE_ERR_* code; tfhe_last_error() gives the message for the last failure on that context.
 *
 * Data formats (all Torus32 = int32_t, wrapping two's-complement arithmetic):
 *   LWE sample      : int32[n+1]          = a[0..n-1], b            (lwe.jl:21-29)
 *   extracted sample: int32[k*N+1]        = a[0..kN-1], b           (tlwe.jl:55-59)
 *   bootstrapping key, canonical Int32 form:
 *       int32 [n][l][k+1][k+1][N]  = key[i].samples[p, j].a[c]      (bootstrap.jl:1-16, tgsw.jl:25-42)
 *       i.e. the inverse_transform of the spectra the reference stores (bootstrap.jl:12-14)
 *   bootstrapping key, the reference's stored form:
 *       complex128 [n][l][k+1][k+1][N/2] (re,im interleaved)        (polynomials.jl:12-14,106-112)
 *   keyswitch key   : int32 [kN][t][base-1][n+1] = key[h, j, i]     (keyswitch.jl:7-42; h fastest in Julia)
 *   MK sample       : int32[P*n+1]        = a[:,0], a[:,1], ..., b  (mk_internals.jl:6-18)
 *   MK bootstrap key: int32 [P][n][2*l*P + 2*l][N], per (party i, bit j) the polys
 *       x[l][P], y[l][P], c0[l], c1[l]                              (mk_internals.jl:243-271,442-461)
 *   MK keyswitch key: P single-key keyswitch keys back to back      (mk_api.jl:83-101)
 *
 * Ownership: the caller owns every host buffer for the duration of the call only; the library
 * copies keys to the device(s) at load time and owns all device memory.  A context made by
 * tfhe_ctx_create is bound to one device; one made by tfhe_ctx_create_multi fans every host-buffer
 * batch call out over its devices on library-owned threads and streams (keys replicated at load,
 * contiguous rotation-balanced shards, results written straight into the caller's buffer).
 *
 * Threading.  The reference is single-threaded and non-re-entrant (global transform plans with shared scratch,
 * polynomials.jl:80-103).  Here distinct contexts are independent, and calls on ONE context must not overlap: the thread
 * inside an entry point owns the context until that call returns, and a call made meanwhile from another thread fails with
 * TFHE_ERR_STATE (tfhe_last_error then tells that thread why) instead of racing on the context's workspaces.  Give every
 * host thread its own context (keys are per context), or serialise the calls (the Julia binding holds a ReentrantLock).
 * Two entry points are exempt and may be called from any thread at any time: tfhe_ctx_synchronize and tfhe_gates_batch_wait
 * (which then waits as tfhe_ctx_synchronize does) — what a finalizer needs before it frees the buffers of a submitted batch.
 * The asynchronous forms (tfhe_gates_batch_submit, tfhe_gates_batch_dev, tfhe_gates_level) return while the device works;
 * only their host side is a "call" in this sense.
 *
 * Parameter sets.  The reference validates nothing (SchemeParameters is a positional struct, tlwe_mask_size a free keyword:
 * api.jl:4-21,30,55); decode_message needs 2N to be a power of two (numeric-functions.jl:28-33) and the transform plans an even
 * length (polynomials.jl:51,69).  tfhe_ctx_create accepts EVERY such set: N any power of two from 2 to 8192, any tlwe_mask_size
 * k, any bs_decomp_length l with l * bs_log2_base <= 32, any lwe_size, any keyswitch base / length with t * gamma <= 31;
 * multi-key: any N, any number of parties, any l (k = 1 as the reference hard-wires it, mk_internals.jl:89-91,129-131).
 * What runs where (every path gives the same words):
 *   tuned kernels      N = 1024 with k = 1 and ANY l (instantiated for the shipped l = 2 and 3; the one- and two-waves-per-rotation
 *                      kernels also exist with l as a run-time value and serve every other l at the same speed); N = 1024 with
 *                      k = 2 and l = 2 or 3; N = 2048 with k = 1 and l = 3 (BASELINE config 4b); N = 512 with k = 1 and any l (the
 *                      same design with four points per lane); multi-key N = 1024: the shipped 2- / 4- / 8-party sets
 *   general kernels    everything else at N = 1024 / 2048 with k <= 4 (blind_rotate_kernel_general, ~3.5 x slower than tuned),
 *                      multi-key at N = 1024 with up to 8 parties and l <= 8 (mk_blind_rotate_kernel_general)
 *   any-N kernels      every other set (N other than 512 / 1024 / 2048, N = 512 with k > 1, k > 4, multi-key with N other than 1024, more than 8 parties or
 *                      l > 8): csrc/kernels_anyn.hpp, one workgroup per rotation, mixed-radix transforms in LDS — correct, untuned
 *   keyswitch          int8 MFMA kernel for base 4 / t = 8 (k N a multiple of 128), tiled integer kernel for base 4 / t a
 *                      multiple of 4 (k N <= 2048), gather kernel for every other base, length and size (single- and multi-key)
 * Refused: N that is no power of two (TFHE_ERR_INVALID_ARG), N > 8192 (TFHE_ERR_UNSUPPORTED: one polynomial's transform no
 * longer fits a compute unit's 160 KB of LDS — and the Float64 transform, here as in the reference, has no rounding margin
 * left there), multi-key with tlwe_mask_size != 1 (TFHE_ERR_UNSUPPORTED: as the reference).
 *
 * Exactness domain.  Like the reference (polynomials.jl:106-132) the external product goes through a Float64 transform and
 * ONE rounding per output coefficient (polynomials.jl:115-116: round(Int64, x), low 32 bits).  Every result word is the exact
 * negacyclic product mod 2^32 as long as (i) the transform's error stays below 1/2 — tfhe_last_rounding_margin measures the
 * largest distance of a pre-rounding value from an integer; 0.03 - 0.08 at the shipped sets, growing with N, l and
 * bs_log2_base — and (ii) every pre-rounding value v satisfies |v| < 2^51: the kernels round with the 1.5 * 2^52 trick, exact
 * there; the reference's round(Int64, .) is defined up to 2^63 but has lost integer precision long before.  With a real key
 * (uniform words) |v| is around sqrt((k+1) l N) * 2^(bs_log2_base + 29) = 2^44 at the 80-bit set; only a caller-supplied
 * "key" whose words all share one sign at magnitude 2^31 reaches 2^52 (k = 1, l = 2, beta = 10, N = 1024): outside the domain,
 * no guarantee there from this engine or from the reference (tests/test_any_params.py::test_worst_case_magnitude_key feeds
 * such a key and records what comes out: with every word -2^31 the values are multiples of 2^31, which Float64 still holds
 * exactly, and both still return the exact product; full-range random words with the extremes planted are exact as any key).
 */
#ifndef TFHE_MI355X_H
#define TFHE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFHE_MI355X_ABI_VERSION 7

/* Scheme parameters — the fields of SchemeParameters the hot path reads (api.jl:4-21). */
typedef struct tfhe_params {
    int32_t n;            /* lwe_size                                  api.jl:6   */
    int32_t N;            /* tlwe_polynomial_degree (power of two)     api.jl:9   */
    int32_t k;            /* tlwe_mask_size                            api.jl:10  */
    int32_t bs_l;         /* bs_decomp_length                          api.jl:12  */
    int32_t bs_log2_base; /* bs_log2_base                              api.jl:13  */
    int32_t ks_t;         /* ks_decomp_length                          api.jl:16  */
    int32_t ks_log2_base; /* ks_log2_base                              api.jl:17  */
    int32_t parties;      /* max_parties (1 = single key)              api.jl:20  */
} tfhe_params;

typedef struct tfhe_ctx tfhe_ctx;

/* error codes */
enum {
    TFHE_OK = 0,
    TFHE_ERR_INVALID_ARG = 1,  /* NULL pointer, bad size, bad opcode                       */
    TFHE_ERR_UNSUPPORTED = 2,  /* parameter set outside what the kernels are built for      */
    TFHE_ERR_NO_KEY = 3,       /* a key needed by the call has not been loaded              */
    TFHE_ERR_DEVICE = 4,       /* HIP runtime error (no device, allocation, launch, ...)    */
    TFHE_ERR_STATE = 5,        /* call not possible in the context's state: wrong kind of context, overlapping call
                                  from another thread, option after the key it decides about; unexpected C++ exception */
    TFHE_ERR_NOMEM = 6         /* host memory exhausted inside the library (std::bad_alloc); ABI v7                    */
};

/* Gate opcodes for tfhe_gates_batch — one per exported gate_* function (TFHE.jl:34-46). */
enum {
    TFHE_GATE_NAND = 0,    /* gate_nand    gates.jl:15-18   */
    TFHE_GATE_OR = 1,      /* gate_or      gates.jl:27-30   */
    TFHE_GATE_AND = 2,     /* gate_and     gates.jl:39-42   */
    TFHE_GATE_XOR = 3,     /* gate_xor     gates.jl:51-54   */
    TFHE_GATE_XNOR = 4,    /* gate_xnor    gates.jl:63-66   */
    TFHE_GATE_NOT = 5,     /* gate_not     gates.jl:76-79   (no bootstrap) */
    TFHE_GATE_NOR = 6,     /* gate_nor     gates.jl:102-105 */
    TFHE_GATE_ANDNY = 7,   /* gate_andny   gates.jl:114-117 */
    TFHE_GATE_ANDYN = 8,   /* gate_andyn   gates.jl:126-129 */
    TFHE_GATE_ORNY = 9,    /* gate_orny    gates.jl:138-141 */
    TFHE_GATE_ORYN = 10,   /* gate_oryn    gates.jl:150-153 */
    TFHE_GATE_MUX = 11,    /* gate_mux     gates.jl:163-177 (2 blind rotations + 1 keyswitch) */
    TFHE_GATE_CONST0 = 12, /* gate_constant(ck, false)  gates.jl:91-93 */
    TFHE_GATE_CONST1 = 13, /* gate_constant(ck, true)   gates.jl:91-93 */
    TFHE_GATE_COPY = 14,   /* identity (circuit plumbing; no reference counterpart needed) */
    TFHE_GATE__COUNT = 15
};

/* ---- library / context ------------------------------------------------------------------- */

/* ABI version of the loaded library (== TFHE_MI355X_ABI_VERSION it was built with).  NEGATIVE: a development build
 * (-DTFHE_EXPERIMENT, csrc/experiment.hpp: in-kernel stamps, environment-variable overrides) that a binding must refuse
 * unless its user asked for it. */
int32_t tfhe_abi_version(void);

/* Number of HIP devices visible to this process (<0: runtime error). */
int32_t tfhe_device_count(void);

/* Replaces the implicit construction of TGswParams / KeyswitchParameters / LweParams from
 * SchemeParameters (api.jl:72-82, tgsw.jl:8-21) and validates what the reference does not
 * (api.jl:4-21 has no checks): N a power of two (2 .. 8192), bs_l*bs_log2_base <= 32, ks_t*ks_log2_base <= 31. */
int32_t tfhe_ctx_create(const tfhe_params *params, int32_t device_id, tfhe_ctx **out_ctx);
void tfhe_ctx_destroy(tfhe_ctx *ctx);

/* Multi-device context (ctx_create(params, device_ids[], n_dev)): the analogue of Julia's
 * `gate_xor.(cloud_key, c1, c2)` broadcast (docs/src/manual.md:28-35) spread over all the GPUs of a node.
 * One device context per entry of device_ids[] (an id may repeat: two contexts then share that GPU on separate
 * streams).  tfhe_load_* / tfhe_mk_load_* replicate the key to every device; tfhe_gates_batch,
 * tfhe_bootstrap_batch, tfhe_keyswitch_batch and tfhe_mk_gate_nand_batch split the batch into contiguous shards
 * (gates: balanced by blind-rotation count, MUX = 2) that run concurrently and write into the caller's buffers.
 * There is no communication between devices: the gates of a batch are independent (gates.jl).
 * tfhe_gates_batch_dev needs n_dev == 1 (a device pointer belongs to one device).  tfhe_gates_batch_submit gives every
 * device its shard as a submit of its own, so each keeps two batches in flight.  The wire table (tfhe_wires_*) is
 * replicated on every device; tfhe_gates_level runs a level of fewer than "level_split_min" blind rotations (option,
 * default 16 per compute unit of the first device: 4096 on an MI355X) on the first device and shards a wider one over all of them.  The context tracks which replicas hold each
 * wire's current value: a device fetches the operand rows it is about to read and does not have — device to device
 * (hipMemcpyPeerAsync; peer access is switched on at creation wherever hipDeviceCanAccessPeer allows) or through pinned host
 * memory where it does not (option "level_exchange": 0 = by peer access, 1 / 2 force either path) — on the devices' own streams,
 * ordered by events: like the one-device call, tfhe_gates_level returns when the level is queued.  Rows nobody else reads
 * never travel; tfhe_wires_download / _gather bring the first device up to date for what they read.  Results are
 * bit-identical to a one-device context. */
int32_t tfhe_ctx_create_multi(const tfhe_params *params, const int32_t *device_ids, int32_t n_dev,
                              tfhe_ctx **out_ctx);
/* Number of device contexts behind ctx (1 for tfhe_ctx_create). */
int32_t tfhe_ctx_device_count(const tfhe_ctx *ctx);
/* The sharding rule of a multi-device tfhe_gates_batch: bounds[r] .. bounds[r+1] (r < shards) are the gates of
 * shard r; contiguous, balanced by blind rotations (MUX = 2, NOT / CONSTANT / COPY = 0; gates.jl:76-93,163-177).
 * opcodes may be NULL (all gates cost one rotation).  bounds: int64 [shards + 1]. */
int32_t tfhe_shard_bounds(const uint8_t *opcodes, int64_t B, int32_t shards, int64_t *bounds);

/* Message of the last error on ctx (ctx == NULL: last error of a failed tfhe_ctx_create). */
const char *tfhe_last_error(const tfhe_ctx *ctx);

/* Copies the parameters the context was created with. */
int32_t tfhe_ctx_params(const tfhe_ctx *ctx, tfhe_params *out);

/* Blocks until everything queued on ctx so far has completed: its own stream, its second stream (tfhe_gates_batch_submit),
 * every device of a multi-device context.  Takes no ownership of the context and touches none of its state: callable from ANY
 * thread, also while another thread is inside a call on ctx (ABI v7).  For code that must release the host buffers of a submitted
 * batch without being the context's caller — a garbage collector's finalizer, an error path: after it returns TFHE_OK no DMA
 * of an earlier submit is still reading or writing them. */
int32_t tfhe_ctx_synchronize(tfhe_ctx *ctx);

/* ---- keys (replace CloudKey's object graphs, api.jl:111-127) -------------------------------- */

/* BootstrapKey (bootstrap.jl:1-16) from the canonical Int32 form [n][l][k+1][k+1][N].
 * The engine forward-transforms it on the device into its own spectrum-domain layout
 * (the analogue of `forward_transform.(bk)`, bootstrap.jl:12).  Any Int32 words are accepted; results are the exact product
 * within the exactness domain stated at the top of this file (|pre-rounding value| < 2^51: every real key). */
int32_t tfhe_load_bootstrap_key_i32(tfhe_ctx *ctx, const int32_t *bk);

/* BootstrapKey from the reference's stored spectra, complex128 [n][l][k+1][k+1][N/2] as produced by
 * polynomials.jl:106-112.  The engine's spectrum domain is the reference's (same fold, twist and
 * transform sign), so loading is a permutation into the engine's order plus the 1/M scaling. */
int32_t tfhe_load_bootstrap_key_c128(tfhe_ctx *ctx, const double *bk_spectra);

/* KeyswitchKey (keyswitch.jl:7-42), Int32 [kN][t][base-1][n+1]. */
int32_t tfhe_load_keyswitch_key(tfhe_ctx *ctx, const int32_t *ks);

/* Generates the cloud key ON THE DEVICE and loads it: BootstrapKey = tgsw_encrypt(s_i) for every bit of the LWE key
 * (bootstrap.jl:6-15, tgsw.jl:52-88, tlwe.jl:63-73) and KeyswitchKey (keyswitch.jl:14-41, lwe.jl:49-55) — the work
 * CloudKey(rng, secret_key) does on the host (api.jl:111-127).  lwe_key: [n] words 0/1 (SecretKey.key, lwe.jl:11-17);
 * tlwe_key: [k][N] words 0/1 (TLweKey, tlwe.jl:11-21); the noise parameters are bs_noise_stddev / ks_noise_stddev of
 * SchemeParameters (api.jl:4-21).  Randomness is Philox4x32-10 (csrc/kernels_keygen.hpp documents the streams; the
 * reference's MersenneTwister stream is not reproduced) keyed by `seed`, SIX 32-bit words: seed[0..1] key the mask
 * words, which the cloud key publishes anyway; seed[2..5] are 128 bits that key the noise and are AS SECRET AS THE SECRET
 * KEY (with them every noise term can be subtracted and the keys solved for): draw them from a cryptographic source, keep
 * them with the secret key or discard them, never with the cloud key.  The call uploads the secret key bits to the
 * device for its duration (the context therefore sees the secret key); the scratch copies and the raw noise are zeroed
 * before their memory is released.  bk_out / ks_out (either may be NULL) receive the canonical Int32 arrays,
 * [n][l][k+1][k+1][N] and [kN][t][base-1][n+1], e.g. to serialise the key.
 * Single-key contexts; a multi-device context generates on its first device and replicates through a host copy. */
int32_t tfhe_keygen_cloud_key(tfhe_ctx *ctx, const int32_t *lwe_key, const int32_t *tlwe_key, double bs_noise_stddev,
                              double ks_noise_stddev, const uint32_t *seed /* [6] */, int32_t *bk_out, int32_t *ks_out);

/* ---- host buffers ---------------------------------------------------------------------------- */

/* Page-locked host memory for the operands and results of the host-buffer batch calls below.  Any host pointer works
 * with them; from pageable memory the runtime stages every copy through its own pinned bounce buffers at roughly half
 * the PCIe rate, from memory allocated here the copies are single DMA transfers (8.2 MB per operand of a 4096-gate
 * batch at the 80-bit set).  Not tied to a context; free with tfhe_host_free. */
int32_t tfhe_host_alloc(size_t bytes, void **out_ptr);
void tfhe_host_free(void *ptr);

/* ---- the hot path --------------------------------------------------------------------------- */

/* B independent gates: out[g] = gate_<opcode[g]>(ck, in0[g], in1[g], in2[g])   (gates.jl:15-177).
 * in0/in1/in2/out: host int32 [B][n+1].  in1 may be NULL if no opcode reads a 2nd operand,
 * in2 may be NULL if no opcode is MUX. */
int32_t tfhe_gates_batch(tfhe_ctx *ctx, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1,
                         const int32_t *in2, int32_t *out, int64_t B);

/* The streaming form of tfhe_gates_batch, for a caller that feeds batch after batch from host memory: submit enqueues the
 * upload, the kernels and the download of one batch and returns; wait blocks until that batch's `out` is complete.  A
 * context runs up to two submitted batches at a time, on two streams taken in turn, so that the upload of batch i + 1
 * runs under the kernels of batch i and the download of batch i under the kernels of batch i + 1 (the PCIe time of
 * the synchronous call — ~0.7 ms of a 12 ms 4096-gate batch — disappears from the steady state); submitting a third
 * batch first waits for the oldest.  in0/in1/in2/out must stay valid and untouched until the batch has been waited
 * for, and should come from tfhe_host_alloc: the runtime stages copies from pageable memory synchronously, which
 * overlaps nothing.  `opcodes` is read before submit returns.  *ticket identifies the batch for tfhe_gates_batch_wait
 * (waiting twice, or for a batch a later submit already displaced, is harmless).  On a multi-device context every device
 * takes its shard as a submit of its own.  Contexts that cannot run two batches side by side (multi-key, measure_margin)
 * complete the batch inside submit.  The timing queries below see only the batches that ran on the context's own stream
 * (every other one). */
int32_t tfhe_gates_batch_submit(tfhe_ctx *ctx, const uint8_t *opcodes, const int32_t *in0, const int32_t *in1,
                                const int32_t *in2, int32_t *out, int64_t B, int32_t *ticket);
/* wait: callable from any thread (ABI v7).  From the thread that submitted, it waits for the ticket and releases its slot.  From any
 * other thread it never takes the context — it cannot make the submitting thread's calls fail — and waits for everything queued
 * on ctx (tfhe_ctx_synchronize); the slot is released by the submitting thread's next submit or wait. */
int32_t tfhe_gates_batch_wait(tfhe_ctx *ctx, int32_t ticket);

/* Same with DEVICE pointers for in0/in1/in2/out (opcodes stay a host array) on HIP stream `stream`
 * (a hipStream_t, NULL = the context's own stream).  Asynchronous with respect to the host except
 * for the upload of the B opcode bytes; results are ordered on `stream`.  The context's workspaces are shared by
 * all calls: every batch call records a context-owned event at its end and the next call makes its own stream wait
 * for it (device-side ordering; the host does not block and no handle of the caller's stream is kept). */
int32_t tfhe_gates_batch_dev(tfhe_ctx *ctx, const uint8_t *opcodes, const int32_t *d_in0,
                             const int32_t *d_in1, const int32_t *d_in2, int32_t *d_out, int64_t B,
                             void *stream);

/* ---- levelised circuits on device-resident ciphertexts ----------------------------------------------
 * The caller side of the path (examples/tutorial.jl:42-62 is a 16-deep XNOR->MUX chain followed by 16
 * parallel MUXes): ciphertexts stay in a device-resident table of `num_wires` LWE samples and each call
 * runs ONE level of independent gates addressed by wire index, so nothing crosses PCIe between levels. */

/* (Re)allocates the context's wire table: int32 [num_wires][n+1] on the device (0 frees it). */
int32_t tfhe_wires_alloc(tfhe_ctx *ctx, int64_t num_wires);
/* Copies `count` samples (host int32 [count][n+1]) into / out of wires [first, first+count). */
int32_t tfhe_wires_upload(tfhe_ctx *ctx, int64_t first, int64_t count, const int32_t *host);
int32_t tfhe_wires_download(tfhe_ctx *ctx, int64_t first, int64_t count, int32_t *host);
/* host[i] = wire[wires[i]] for i < count: any set of wires in ONE device gather + ONE copy (the outputs of a circuit). */
int32_t tfhe_wires_gather(tfhe_ctx *ctx, const int32_t *wires, int64_t count, int32_t *host);
/* wire[out[g]] = gate_<opcode[g]>(ck, wire[a[g]], wire[b[g]], wire[c[g]]) for g < B (host index arrays; b / c
 * may be NULL if no opcode reads them).  A level must not read a wire it writes, nor write a wire twice
 * (TFHE_ERR_INVALID_ARG).  Asynchronous: returns when the level is queued on the context's stream (multi-device context: on
 * every participating device's stream, the exchange of operand rows between devices included); ordered with later calls. */
int32_t tfhe_gates_level(tfhe_ctx *ctx, const uint8_t *opcodes, const int32_t *a, const int32_t *b,
                         const int32_t *c, const int32_t *out, int64_t B);

/* bootstrap(bk, ks, mu, x) (bootstrap.jl:92-95) if with_keyswitch != 0, else
 * bootstrap_wo_keyswitch(bk, mu, x) (bootstrap.jl:69-82).
 * in: host int32 [B][n+1]; out: host int32 [B][n+1] or [B][k*N+1]. */
int32_t tfhe_bootstrap_batch(tfhe_ctx *ctx, int32_t mu, const int32_t *in, int32_t *out, int64_t B,
                             int32_t with_keyswitch);

/* keyswitch(ks, sample) (keyswitch.jl:45-80). in: host int32 [B][k*N+1]; out: host int32 [B][n+1]. */
int32_t tfhe_keyswitch_batch(tfhe_ctx *ctx, const int32_t *in, int32_t *out, int64_t B);

/* ---- multi-key (mk_gates.jl:7-12, mk_internals.jl:348-411,464-515) --------------------------- */

/* MKBootstrapKey from Int32 [P][n][2*l*P + 2*l][N] (see top of file); 2 <= P <= the context's `parties` (mk_api.jl:94). */
int32_t tfhe_mk_load_bootstrap_key_i32(tfhe_ctx *ctx, const int32_t *bk, int32_t parties);
/* The same key in the form the reference stores it (MKBootstrapKey.key[j, i] :: MKTransformedTGswExpSample,
 * mk_internals.jl:274-288,442-461): complex128 [P][n][2*l*P + 2*l][N/2] spectra of polynomials.jl:106-112.  Loading
 * is a permutation into the engine's order plus the 1/M scaling: nothing is re-transformed or rounded. */
int32_t tfhe_mk_load_bootstrap_key_c128(tfhe_ctx *ctx, const double *bk_spectra, int32_t parties);
/* MKBootstrapKey built ON THE DEVICE from what the parties publish: RGSW.Expand (mk_tgsw_expand, mk_internals.jl:304-345)
 * of every party's uni-encrypted key bits against all public keys, then the forward transform (MKBootstrapKey,
 * :442-461).  pub_b: Int32 [P][l][N] (PublicKey.b, :116-139); c0, c1, d0, d1, f0, f1: Int32 [P][n][l][N]
 * (MKTGswUESample of bit j of party i, :185-227).  The host only decomposes the public-key differences; the
 * P (P-1) n l^2 polynomial products run on the GPU (exact, like the external product).  If expanded_out is not NULL it
 * receives the expanded key in the Int32 layout tfhe_mk_load_bootstrap_key_i32 takes ([P][n][2 l P + 2 l][N]). */
int32_t tfhe_mk_expand_load_bootstrap_key(tfhe_ctx *ctx, int32_t parties, const int32_t *pub_b, const int32_t *c0,
                                          const int32_t *c1, const int32_t *d0, const int32_t *d1, const int32_t *f0,
                                          const int32_t *f1, int32_t *expanded_out);
/* P single-key KeyswitchKeys back to back, each [N][t][base-1][n+1]; any base and length (mk_internals.jl:397-411 calls the
 * single-key keyswitch per party). */
int32_t tfhe_mk_load_keyswitch_key(tfhe_ctx *ctx, const int32_t *ks, int32_t parties);
/* out[g] = mk_gate_nand(ck, in0[g], in1[g]); all host int32 [B][P*n+1]. */
int32_t tfhe_mk_gate_nand_batch(tfhe_ctx *ctx, const int32_t *in0, const int32_t *in1, int32_t *out,
                                int64_t B);

/* ---- measurement ---------------------------------------------------------------------------- */

/* Timing of the most recent batch call on ctx, from HIP events recorded on the stream the kernels
 * were launched on.  which: 0 = blind-rotate kernel(s), 1 = keyswitch kernel(s), 2 = whole batch
 * (prologue .. last kernel, device side).  Blocks until those kernels have finished.  After a host-buffer call that ran as
 * two halves on two streams (tfhe_gates_batch from "pipeline_min" gates up; default 16 per compute unit: 4096 on an MI355X): from the start of the phase on the stream that
 * started first to the later of the two streams' ends. */
int32_t tfhe_last_timing_ms(tfhe_ctx *ctx, int32_t which, float *ms);

/* The same timing of up to the last 32 batch calls on a one-device ctx, oldest first, read in ONE go after the calls: a
 * caller timing a sequence of asynchronous calls need not synchronise (and so serialise its host work with the device)
 * after each one.  ms: room for max_calls floats; *n_out = entries written (<= max_calls, <= 32, <= calls made).  A call
 * that failed part-way is not counted.  The history is per stream: after tfhe_gates_batch_submit it holds the batches that
 * took the context's own stream (every other one); a blocking host-buffer call that ran as two halves contributes the half
 * on the own stream (tfhe_last_timing_ms spans both). */
int32_t tfhe_timing_history_ms(tfhe_ctx *ctx, int32_t which, float *ms, int32_t max_calls, int32_t *n_out);

/* Number of device contexts that took part in the most recent batch / level call (1 on a one-device context; on a
 * multi-device context: how many devices the call was sharded over — a narrow circuit level runs on the first one).  ABI v5. */
int32_t tfhe_last_device_count(const tfhe_ctx *ctx);

/* Number of blind rotations the most recent batch call executed (MUX counts 2). */
int64_t tfhe_last_rotation_count(const tfhe_ctx *ctx);

/* Name of the blind-rotate kernel instantiation the most recent batch call launched (e.g.
 * "blind_rotate_kernel_v3<2,16>"); valid until the next call on ctx. */
const char *tfhe_last_kernel_name(const tfhe_ctx *ctx);

/* Exactness evidence for the Float64 transform: after tfhe_set_option(ctx, "measure_margin", 1), batch calls run
 * the DIAG instantiation of whichever blind-rotate kernel the dispatcher selects (every kernel that rounds has one:
 * single key N = 1024 / 2048, k = 1 / 2, two-wave, multi-key 2-party and any-party) and record, per blind rotation,
 * the largest distance of any pre-rounding value from an integer (polynomials.jl:115-116 rounds; a flipped rounding
 * needs 0.5).  Returns the maximum over the last call. */
int32_t tfhe_last_rounding_margin(tfhe_ctx *ctx, double *worst);

/* Same DIAG run: the shader clock the blind-rotate kernel actually held, in MHz = s_memtime ticks / s_memrealtime
 * ticks x 100 MHz around the kernel body, median over the workgroups that ran for at least 10 us (what FP64-issue roofline
 * figures are priced at).  TFHE_ERR_STATE if no workgroup ran that long: the 100 MHz counter is too coarse for a shorter one.
 * A kernel of a few tens of microseconds right after the device woke up reads the clock of the DVFS ramp, not the sustained
 * one: price rooflines with the reading of a launch that lasts milliseconds. */
int32_t tfhe_last_kernel_clock_mhz(tfhe_ctx *ctx, double *mhz);

/* Options by name.  NONE of them changes a result word: they choose among kernels that compute the same thing (the tests
 * force every kernel through them and compare with the oracle), switch diagnostics on, or move a dispatch threshold.  Defaults
 * are what the measurements of DESIGN.md chose; thresholds given per compute unit scale with the device (256 CUs on an MI355X).
 *
 *   what runs                  "measure_margin" 0|1       the DIAG instantiation of the chosen kernel: rounding margin + in-kernel clock
 *                              "timing_events"  1|0       0: no per-phase HIP events (each costs the stream ~5 us; tfhe_last_timing_ms then reports nothing)
 *   host-buffer pipeline       "pipeline_min"   n         tfhe_gates_batch from n gates up runs as two halves on two streams (default 16 per CU; < 0 never)
 *   multi-device context       "level_split_min" n        levels of at least n rotations are sharded over the devices (default 16 per CU; < 0 never)
 *                              "level_exchange" 0|1|2     rows between devices: by peer access | device to device | pinned host staging
 *   kernel by batch size       "br_tiny"  n               up to n rotations: 4 l waves per rotation (h2); default one per CU, -1 never
 *    (N = 1024, k = 1)         "br_small" n               up to n rotations: two waves per rotation (w2); default 4 per CU, -1 never
 *                              "br_split" 1|0             a batch above what the chip holds sends its last partly filled round to the two-wave kernel
 *                              "w2_rw" 0|1|2, "v3_rw" 0|1|4   rotations per workgroup in lockstep (0: by batch size)
 *                              "br_prio_pct" 0..100       a wave lowers its issue priority over this share of its steps (default 90)
 *                              "br_rt_l" 0|1              l = 2 / 3 on the run-time-l instantiations too (a comparison switch)
 *   other shapes               "k2_w3" -1|0|1, "k2_rw" 0|1|7       k = 2: three waves per rotation (by size | never | always), lockstep groups
 *                              "n2048_rw" 0|1|2           N = 2048: rotations per workgroup
 *                              "n512_w2" -1|0|1, "n512_rw" 0|1|4   N = 512
 *                              "br_general" 0|1           every single-key rotation on blind_rotate_kernel_general (cross-check)
 *                              "br_anyn" 0|1, "anyn_spec" -1|0|1   the any-N kernel where a tuned one exists (decides a KEY LAYOUT: choose before
 *                                                         loading the bootstrapping key, TFHE_ERR_STATE otherwise); its spectrum accumulators in LDS / global memory
 *   multi-key                  "mk_rw" 0|1|2, "mk_general" 0|1, "mkg_variant" 0|1, "mkg_rw" 0|1|2|4, "mkg_acc" -1|0|1
 *   keyswitch                  "ks_variant" 4|3|1         int8 MFMA | tiled integer | gather (decides a KEY LAYOUT: before loading the keyswitch key)
 *                              "ks_slices" 1|2|4          K-split of the MFMA kernel for large batches
 *   read-only (tfhe_get_option) "exact_domain", "exact_bound_log2_x1000", "exact_margin_x1e6" (below), "peer_pairs" (multi-device context:
 *                              ordered pairs of different device contexts that copy device to device)
 *   tests                      "debug_fail_alloc_after" n (ctx may be NULL; process-wide): see tfhe_get_option */
int32_t tfhe_set_option(tfhe_ctx *ctx, const char *name, int64_t value);
/* Read-only names of tfhe_get_option (ABI v7), decided by tfhe_ctx_create from the parameter set alone — is it inside what a
 * Float64 transform computes exactly ("Exactness domain" above)?
 *   "exact_domain"            2 = every result word is the exact product for ANY Int32 key words: the worst-case pre-rounding
 *                                 magnitude np N 2^(beta-1) 2^31 (np = (k+1) l products per output; multi-key (P+1) l) is below 2^51
 *                                 and the predicted rounding margin below 1/4  (tfhe_parameters_128, the multi-key sets);
 *                             1 = the same for every REAL key (uniform mask words: 8 x rms magnitude below 2^51), but not for an
 *                                 adversarial one  (tfhe_parameters_80: its all-keys bound is exactly 2^52);
 *                             0 = outside: the predicted margin reaches 1/4 (or the magnitude 2^51).  The engine still computes —
 *                                 the reference does too and warns (polynomials.jl:135-144) — but a word may differ from the exact
 *                                 product; tfhe_last_rounding_margin measures.  The Python and Julia constructors warn once.
 *   "exact_bound_log2_x1000"  1000 x log2 of that worst-case magnitude
 *   "exact_margin_x1e6"       10^6 x the predicted margin = 4.5 x 2^-53 x rms x max(log2(N/2), 4), rms = sqrt(np N) 2^(beta+32) / 12; every
 *                             measured margin of rounds 2-6 lies between 0.31 and 0.90 of it
 * "debug_fail_alloc_after" (set / get, ctx may be NULL, process-wide, default 0 = off): the n-th allocation checkpoint inside the
 * library from now on throws std::bad_alloc — how the tests show that TFHE_ERR_NOMEM comes back and the context survives. */
/* The current value of an option (ABI v6): a caller that changes one for a while can put it back.  "br_anyn" (the any-N kernel
 * where a tuned one exists; like "ks_variant" it decides a key layout and must be chosen before the bootstrapping key is
 * loaded), "anyn_spec", "level_exchange", "k2_w3", "n512_rw", "n512_w2" (which N = 512 / k = 2 kernel a batch size takes) are new in v6. */
int32_t tfhe_get_option(tfhe_ctx *ctx, const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* TFHE_MI355X_H */
