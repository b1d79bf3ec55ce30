/*
 * tfhe_oracle.c — CPU restatement of TFHE.jl's gate-bootstrapping hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and there
 * only as the checker / the timed CPU baseline.  The shipped path is the HIP library in
 * tfhe.jl_amd/csrc (C ABI: include/tfhe_mi355x.h) and fails loudly without it.
 *
 * PARITY STATUS: "parity unpinned" at the ciphertext-word level.  The reference (Julia, with the
 * un-vendored dependencies DarkIntegers ~0.1.0 and FFTW.jl, Project.toml:6-20) cannot run in the
 * build container and its own tests (test/runtests.jl:26-100) pin only decrypted Booleans — there
 * are no golden vectors, KATs or fixtures at the Int32 level.  This restatement is therefore
 * anchored on (i) the reference's truth tables under valid keys, (ii) its closed-form constants
 * (gadget / offset values, encode/decode), (iii) agreement between two independent product
 * back-ends written here: the reference's folded N/2-point Float64 FFT + round
 * (polynomials.jl:106-132) and the exact negacyclic product in Z_{2^32}[X]/(X^N+1).
 *
 * Each function cites the reference lines it follows (paths relative to /root/reference).
 * DarkIntegers semantics used (source not in /root/reference; call sites cited):
 *   mul_by_monomial(p, s) = X^s * p mod (X^N + 1) for any integer s (tlwe.jl:92-93,
 *   bootstrap.jl:21,54, polynomials.jl:34); Polynomial +/- Polynomial is element-wise wrapping.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_N 8192
#define ORC_MAX_L 32        /* single key: any l with l * beta <= 32 */
#define ORC_MAX_L_MK 32     /* multi-key: the same bound (the digit buffers are per-thread heap blocks sized by need) */
#define ORC_MAX_K 8
#define ORC_MAX_PARTIES 64

typedef struct {
    int32_t n;        /* lwe_size                 api.jl:6  */
    int32_t N;        /* tlwe_polynomial_degree   api.jl:9  */
    int32_t k;        /* tlwe_mask_size           api.jl:10 */
    int32_t l;        /* bs_decomp_length         api.jl:12 */
    int32_t log2Bg;   /* bs_log2_base             api.jl:13 */
    int32_t t;        /* ks_decomp_length         api.jl:16 */
    int32_t log2ks;   /* ks_log2_base             api.jl:17 */
    int32_t parties;  /* max_parties              api.jl:20 */
} orc_params;

/* gate opcodes — shared numbering with include/tfhe_mi355x.h */
enum {
    ORC_NAND = 0, ORC_OR = 1, ORC_AND = 2, ORC_XOR = 3, ORC_XNOR = 4, ORC_NOT = 5,
    ORC_NOR = 6, ORC_ANDNY = 7, ORC_ANDYN = 8, ORC_ORNY = 9, ORC_ORYN = 10, ORC_MUX = 11,
    ORC_CONST0 = 12, ORC_CONST1 = 13, ORC_COPY = 14
};

static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t wneg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }
/* Julia's >> on Int32 is arithmetic; gcc's >> on negative int32_t is arithmetic too. */
static inline int32_t asr(int32_t a, int s) { return a >> s; }

/* numeric-functions.jl:42-45 — Torus32(mu) << (32 - log2(space)) */
int32_t orc_encode_message(int32_t mu, int32_t log2_space)
{
    return (int32_t)((uint32_t)mu << (32 - log2_space));
}

/* numeric-functions.jl:31-34 — (phase + 1 << (32-log2-1)) >> (32-log2), wrapping add, arithmetic shift */
int32_t orc_decode_message(int32_t phase, int32_t log2_space)
{
    return asr(wadd(phase, (int32_t)(1u << (32 - log2_space - 1))), 32 - log2_space);
}

static int ilog2(int x) { int r = 0; while ((1 << r) < x) r++; return r; }

/* tgsw.jl:8-21 — gadget values 2^(32 - p*log2Bg), offset = sum(gadget) * Bg/2 as wrapped Int32 */
void orc_tgsw_constants(int32_t l, int32_t log2Bg, int32_t *gadget /*[l]*/, int32_t *offset)
{
    uint32_t sum = 0;
    for (int p = 1; p <= l; p++) {
        gadget[p - 1] = (int32_t)(1u << (32 - p * log2Bg));
        sum += (uint32_t)gadget[p - 1];
    }
    *offset = (int32_t)(sum * (1u << (log2Bg - 1)));
}

/* tgsw.jl:99-117 — digit_p(c) = (((c + offset) >> (32 - p*log2Bg)) & (Bg-1)) - Bg/2 */
void orc_decompose(const int32_t *poly, int32_t N, int32_t l, int32_t log2Bg, int32_t *out /*[l][N]*/)
{
    int32_t gadget[ORC_MAX_L], offset;
    orc_tgsw_constants(l, log2Bg, gadget, &offset);
    const int32_t mask = (int32_t)((1u << log2Bg) - 1);
    const int32_t part_offset = (int32_t)(1u << (log2Bg - 1));
    for (int p = 1; p <= l; p++)
        for (int j = 0; j < N; j++)
            out[(p - 1) * N + j] = (asr(wadd(poly[j], offset), 32 - p * log2Bg) & mask) - part_offset;
}

/* DarkIntegers mul_by_monomial: X^s * p mod (X^N+1), any integer s (period 2N). */
void orc_mul_by_monomial(const int32_t *p, int32_t N, int32_t s, int32_t *out)
{
    int32_t sm = s % (2 * N);
    if (sm < 0) sm += 2 * N;
    for (int j = 0; j < N; j++) {
        int32_t idx = j - sm;               /* in (-2N, N) */
        idx %= 2 * N;
        if (idx < 0) idx += 2 * N;          /* in [0, 2N) */
        out[j] = idx < N ? p[idx] : wneg(p[idx - N]);
    }
}

/* exact negacyclic product in Z_{2^32}[X]/(X^N+1): out (+)= a (*) b */
static void negacyclic_mac_exact(const int32_t *a, const int32_t *b, int32_t N, uint32_t *acc)
{
    for (int i = 0; i < N; i++) {
        const uint32_t ai = (uint32_t)a[i];
        if (ai == 0) continue;
        const uint32_t nai = 0u - ai;
        uint32_t *o = acc + i;
        int lim = N - i;
        for (int j = 0; j < lim; j++) o[j] += ai * (uint32_t)b[j];
        o = acc - lim;
        for (int j = lim; j < N; j++) o[j] += nai * (uint32_t)b[j];
    }
}

void orc_negacyclic_mul_exact(const int32_t *a, const int32_t *b, int32_t N, int32_t *out)
{
    uint32_t acc[ORC_MAX_N];
    memset(acc, 0, sizeof(uint32_t) * (size_t)N);
    negacyclic_mac_exact(a, b, N, acc);
    for (int j = 0; j < N; j++) out[j] = (int32_t)acc[j];
}

/* ---------------------------------------------------------------------------------------------
 * The reference's transform (polynomials.jl:44-132): an N/2-point complex FFT of the folded,
 * twisted sequence.  The FFT itself is FFTW in the reference (plan_fft / normalised plan_ifft);
 * here an iterative radix-2 FFT with long-double-derived twiddles.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int N;          /* polynomial length */
    int M;          /* N/2 */
    double *tw_re, *tw_im;      /* exp(-2 pi i j / M), j < M/2 : FFT twiddles            */
    double *co_re, *co_im;      /* exp(-i pi j / N), j < M    : polynomials.jl:53,72     */
    int *rev;
} orc_plan;

static orc_plan g_plans[16];      /* one per polynomial length seen by this process */
static int g_nplans = 0;

/* Must be called (single-threaded) before any transform of length N. */
int orc_init(int32_t N)
{
    for (int i = 0; i < g_nplans; i++) if (g_plans[i].N == N) return 0;
    if (g_nplans >= 16 || N > ORC_MAX_N || N < 2 || (N & (N - 1))) return -1;
    orc_plan *p = &g_plans[g_nplans];
    const int M = N / 2;
    p->N = N; p->M = M;
    p->tw_re = malloc(sizeof(double) * M); p->tw_im = malloc(sizeof(double) * M);
    p->co_re = malloc(sizeof(double) * M); p->co_im = malloc(sizeof(double) * M);
    p->rev = malloc(sizeof(int) * M);
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int j = 0; j < M / 2 + 1 && j < M; j++) {
        p->tw_re[j] = (double)cosl(-2.0L * pi * j / M);
        p->tw_im[j] = (double)sinl(-2.0L * pi * j / M);
    }
    for (int j = 0; j < M; j++) {
        p->co_re[j] = (double)cosl(-pi * j / N);
        p->co_im[j] = (double)sinl(-pi * j / N);
    }
    int lg = ilog2(M);
    for (int j = 0; j < M; j++) {
        int r = 0;
        for (int b = 0; b < lg; b++) if (j & (1 << b)) r |= 1 << (lg - 1 - b);
        p->rev[j] = r;
    }
    g_nplans++;
    return 0;
}

static const orc_plan *get_plan(int N)
{
    for (int i = 0; i < g_nplans; i++) if (g_plans[i].N == N) return &g_plans[i];
    return NULL;
}

/* in-place forward (sign -1) or backward (sign +1, unnormalised) complex FFT of length M */
static void fft_inplace(const orc_plan *p, double *re, double *im, int sign)
{
    const int M = p->M;
    for (int j = 0; j < M; j++) {
        int r = p->rev[j];
        if (r > j) {
            double t = re[j]; re[j] = re[r]; re[r] = t;
            t = im[j]; im[j] = im[r]; im[r] = t;
        }
    }
    for (int len = 2; len <= M; len <<= 1) {
        const int half = len >> 1, step = M / len;
        for (int s = 0; s < M; s += len) {
            for (int j = 0; j < half; j++) {
                const double wr = p->tw_re[j * step];
                const double wi = sign < 0 ? p->tw_im[j * step] : -p->tw_im[j * step];
                const int a = s + j, b = a + half;
                const double xr = re[b] * wr - im[b] * wi;
                const double xi = re[b] * wi + im[b] * wr;
                re[b] = re[a] - xr; im[b] = im[a] - xi;
                re[a] += xr; im[a] += xi;
            }
        }
    }
}

/* polynomials.jl:106-112 — buffer = (c[1:N/2] - im*c[N/2+1:end]) * coeffs ; plan * buffer */
int orc_forward_transform(const int32_t *c, int32_t N, double *out_re, double *out_im)
{
    const orc_plan *p = get_plan(N);
    if (!p) return -1;
    const int M = p->M;
    for (int j = 0; j < M; j++) {
        const double a = (double)c[j], b = -(double)c[j + M];
        out_re[j] = a * p->co_re[j] - b * p->co_im[j];
        out_im[j] = a * p->co_im[j] + b * p->co_re[j];
    }
    fft_inplace(p, out_re, out_im, -1);
    return 0;
}

/* polynomials.jl:115-116 — round(Int64, x) then keep the low 32 bits */
static inline int32_t to_int32(double x)
{
    return (int32_t)(uint32_t)(uint64_t)(int64_t)llround(x);
}

/* polynomials.jl:119-132 — ifft (normalised); conj(.) * coeffs ; real -> 0..N/2-1, imag -> N/2..N-1.
 * If frac_margin != NULL it receives the max distance of any pre-round value from an integer. */
int orc_inverse_transform(const double *in_re, const double *in_im, int32_t N, int32_t *out,
                          double *frac_margin)
{
    const orc_plan *p = get_plan(N);
    if (!p) return -1;
    const int M = p->M;
    double re[ORC_MAX_N / 2], im[ORC_MAX_N / 2];
    memcpy(re, in_re, sizeof(double) * M);
    memcpy(im, in_im, sizeof(double) * M);
    fft_inplace(p, re, im, +1);
    const double inv = 1.0 / M;
    double worst = 0.0;
    for (int j = 0; j < M; j++) {
        const double yr = re[j] * inv, yi = -(im[j] * inv);          /* conj */
        const double zr = yr * p->co_re[j] - yi * p->co_im[j];
        const double zi = yr * p->co_im[j] + yi * p->co_re[j];
        out[j] = to_int32(zr);
        out[j + M] = to_int32(zi);
        if (frac_margin) {
            double d = fabs(zr - nearbyint(zr)); if (d > worst) worst = d;
            d = fabs(zi - nearbyint(zi)); if (d > worst) worst = d;
        }
    }
    if (frac_margin) *frac_margin = worst;
    return 0;
}

/* polynomials.jl:142-144 — transformed_mul */
int orc_negacyclic_mul_fft(const int32_t *a, const int32_t *b, int32_t N, int32_t *out, double *margin)
{
    double ar[ORC_MAX_N / 2], ai[ORC_MAX_N / 2], br[ORC_MAX_N / 2], bi[ORC_MAX_N / 2];
    if (orc_forward_transform(a, N, ar, ai) || orc_forward_transform(b, N, br, bi)) return -1;
    const int M = N / 2;
    for (int j = 0; j < M; j++) {
        const double r = ar[j] * br[j] - ai[j] * bi[j];
        const double i = ar[j] * bi[j] + ai[j] * br[j];
        ar[j] = r; ai[j] = i;
    }
    return orc_inverse_transform(ar, ai, N, out, margin);
}

/* bootstrap.jl:12 (forward_transform.(bk)) with tgsw.jl:120-121, tlwe.jl:96-97:
 * bk_i32 layout [n][l][k+1][k+1][N]  ->  spectra [n][l][k+1][k+1][N/2] (re and im planes). */
int orc_bk_transform(const orc_params *P, const int32_t *bk_i32, double *bk_re, double *bk_im, int64_t npolys)
{
    const int N = P->N, M = N / 2;
    if (!get_plan(N)) return -1;
    int rc = 0;
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < npolys; q++)
        if (orc_forward_transform(bk_i32 + q * N, N, bk_re + q * M, bk_im + q * M)) rc = -1;
    return rc;
}

/* tlwe.jl:55-59 with polynomials.jl:32-35: a'[0] = p[0], a'[m] = -p[N-m]; b' = body[0] */
static void extract_sample(const int32_t *acc /*[(k+1)][N]*/, int k, int N, int32_t *out /*[kN+1]*/)
{
    for (int c = 0; c < k; c++) {
        const int32_t *p = acc + c * N;
        int32_t *o = out + c * N;
        o[0] = p[0];
        for (int m = 1; m < N; m++) o[m] = wneg(p[N - m]);
    }
    out[k * N] = acc[k * N];
}

/* tgsw.jl:125-129 — external product, both back-ends.
 *   temp : (k+1) polys ; result added into acc (mux_rotate's "accum + ...", bootstrap.jl:22).
 *   mode 0: reference-style FFT (bk spectra) ; mode 1: exact integer (bk_i32). */
static int extern_mul_add(const orc_params *P, const int32_t *temp, const double *bkre, const double *bkim,
                          const int32_t *bki32, int mode, int32_t *acc, double *margin)
{
    const int N = P->N, M = N / 2, k1 = P->k + 1, l = P->l;
    int32_t dec[ORC_MAX_L * ORC_MAX_N];
    if (mode == 1) {
        uint32_t out[(ORC_MAX_K + 1) * ORC_MAX_N];
        memset(out, 0, sizeof(uint32_t) * (size_t)(k1 * N));
        for (int j = 0; j < k1; j++) {
            orc_decompose(temp + j * N, N, l, P->log2Bg, dec);
            for (int p = 0; p < l; p++)
                for (int c = 0; c < k1; c++)
                    negacyclic_mac_exact(dec + p * N, bki32 + (((size_t)p * k1 + j) * k1 + c) * N, N, out + c * N);
        }
        for (int j = 0; j < k1 * N; j++) acc[j] = (int32_t)((uint32_t)acc[j] + out[j]);
        return 0;
    }
    double sre[(ORC_MAX_K + 1) * ORC_MAX_N / 2], sim[(ORC_MAX_K + 1) * ORC_MAX_N / 2];
    double dre[ORC_MAX_N / 2], dim[ORC_MAX_N / 2];
    memset(sre, 0, sizeof(double) * (size_t)(k1 * M));
    memset(sim, 0, sizeof(double) * (size_t)(k1 * M));
    for (int j = 0; j < k1; j++) {
        orc_decompose(temp + j * N, N, l, P->log2Bg, dec);                 /* tgsw.jl:126 */
        for (int p = 0; p < l; p++) {
            orc_forward_transform(dec + p * N, N, dre, dim);               /* tgsw.jl:127 */
            for (int c = 0; c < k1; c++) {                                 /* tgsw.jl:128, tlwe.jl:105-111 */
                const size_t off = (((size_t)p * k1 + j) * k1 + c) * M;
                const double *kr = bkre + off, *ki = bkim + off;
                double *orr = sre + c * M, *oi = sim + c * M;
                for (int f = 0; f < M; f++) {
                    orr[f] += dre[f] * kr[f] - dim[f] * ki[f];
                    oi[f] += dre[f] * ki[f] + dim[f] * kr[f];
                }
            }
        }
    }
    int32_t prod[ORC_MAX_N];
    for (int c = 0; c < k1; c++) {
        double mg = 0.0;
        orc_inverse_transform(sre + c * M, sim + c * M, N, prod, margin ? &mg : NULL);  /* tgsw.jl:128 */
        if (margin && mg > *margin) *margin = mg;
        for (int j = 0; j < N; j++) acc[c * N + j] = wadd(acc[c * N + j], prod[j]);
    }
    return 0;
}

/* bootstrap.jl:69-82 + 50-59 + 32-39 + 19-23: modswitch, test vector, blind rotate, extract.
 *   x   : LWE sample [n+1] (a then b)
 *   out : extracted LWE sample [kN+1]
 *   bk_re/bk_im: spectra [n][l][k+1][k+1][N/2] (mode 0) ; bk_i32 (mode 1) */
int orc_bootstrap_wo_keyswitch(const orc_params *P, const double *bk_re, const double *bk_im,
                               const int32_t *bk_i32, int32_t mode, int32_t mu, const int32_t *x,
                               int32_t *out, double *margin)
{
    const int N = P->N, n = P->n, k1 = P->k + 1, l = P->l;
    if (N > ORC_MAX_N || l > ORC_MAX_L || P->k > ORC_MAX_K) return -1;
    if (mode == 0 && !get_plan(N)) return -1;
    const int log2_2N = ilog2(2 * N);
    int32_t acc[(ORC_MAX_K + 1) * ORC_MAX_N], temp[(ORC_MAX_K + 1) * ORC_MAX_N], tv[ORC_MAX_N];
    if (margin) *margin = 0.0;

    const int32_t barb = orc_decode_message(x[n], log2_2N);                 /* bootstrap.jl:75 */
    for (int j = 0; j < N; j++) tv[j] = mu;                                 /* bootstrap.jl:78 */
    memset(acc, 0, sizeof(int32_t) * (size_t)(k1 * N));                     /* tlwe.jl:77-81   */
    orc_mul_by_monomial(tv, N, -barb, acc + (k1 - 1) * N);                  /* bootstrap.jl:54 */

    const size_t per_i = (size_t)l * k1 * k1;
    for (int i = 0; i < n; i++) {                                           /* bootstrap.jl:33 */
        const int32_t bara = orc_decode_message(x[i], log2_2N);             /* bootstrap.jl:74 */
        if (bara == 0) continue;                                            /* bootstrap.jl:34 */
        for (int c = 0; c < k1; c++) {                                      /* bootstrap.jl:21 */
            orc_mul_by_monomial(acc + c * N, N, bara, temp + c * N);
            for (int j = 0; j < N; j++) temp[c * N + j] = wsub(temp[c * N + j], acc[c * N + j]);
        }
        const size_t off = (size_t)i * per_i;
        extern_mul_add(P, temp,
                       bk_re ? bk_re + off * (N / 2) : NULL, bk_im ? bk_im + off * (N / 2) : NULL,
                       bk_i32 ? bk_i32 + off * N : NULL, mode, acc, margin);   /* bootstrap.jl:22 */
    }
    extract_sample(acc, P->k, N, out);                                      /* bootstrap.jl:58 */
    return 0;
}

/* keyswitch.jl:45-80 — ks layout [kN][t][base-1][n+1] (= key[h, j, i], h fastest) */
void orc_keyswitch(const orc_params *P, const int32_t *ks, const int32_t *in /*[kN+1]*/, int32_t *out /*[n+1]*/)
{
    const int n = P->n, kN = P->k * P->N, t = P->t, g = P->log2ks;
    const int base = 1 << g, mask = base - 1;
    memset(out, 0, sizeof(int32_t) * (size_t)n);
    out[n] = in[kN];                                                        /* keyswitch.jl:50 */
    const int32_t prec_offset = (int32_t)(1u << (32 - (1 + g * t)));        /* keyswitch.jl:58 */
    for (int i = 0; i < kN; i++) {
        const int32_t aibar = wadd(in[i], prec_offset);                     /* keyswitch.jl:59 */
        for (int j = 1; j <= t; j++) {
            const int32_t d = asr(aibar, 32 - j * g) & mask;                /* keyswitch.jl:65-67 */
            if (d != 0) {                                                   /* keyswitch.jl:73 */
                const int32_t *row = ks + (((size_t)i * t + (j - 1)) * (base - 1) + (d - 1)) * (size_t)(n + 1);
                for (int w = 0; w <= n; w++) out[w] = wsub(out[w], row[w]); /* keyswitch.jl:74, lwe.jl:71-72 */
            }
        }
    }
}

/* gates.jl prologues: result = (0, const) + sx*x + sy*y, optionally * 2 (XOR/XNOR) */
static int gate_prologue(int op, int *cmu, int *clog, int *sx, int *sy, int *mul2)
{
    *mul2 = 0;
    switch (op) {
    case ORC_NAND:  *cmu = 1;  *clog = 3; *sx = -1; *sy = -1; return 0;    /* gates.jl:16  */
    case ORC_OR:    *cmu = 1;  *clog = 3; *sx = 1;  *sy = 1;  return 0;    /* gates.jl:28  */
    case ORC_AND:   *cmu = -1; *clog = 3; *sx = 1;  *sy = 1;  return 0;    /* gates.jl:40  */
    case ORC_XOR:   *cmu = 1;  *clog = 2; *sx = 1;  *sy = 1;  *mul2 = 1; return 0;   /* gates.jl:52 */
    case ORC_XNOR:  *cmu = -1; *clog = 2; *sx = -1; *sy = -1; *mul2 = 1; return 0;   /* gates.jl:64 */
    case ORC_NOR:   *cmu = -1; *clog = 3; *sx = -1; *sy = -1; return 0;    /* gates.jl:103 */
    case ORC_ANDNY: *cmu = -1; *clog = 3; *sx = -1; *sy = 1;  return 0;    /* gates.jl:115 */
    case ORC_ANDYN: *cmu = -1; *clog = 3; *sx = 1;  *sy = -1; return 0;    /* gates.jl:127 */
    case ORC_ORNY:  *cmu = 1;  *clog = 3; *sx = -1; *sy = 1;  return 0;    /* gates.jl:139 */
    case ORC_ORYN:  *cmu = 1;  *clog = 3; *sx = 1;  *sy = -1; return 0;    /* gates.jl:151 */
    default: return -1;
    }
}

static void affine(int n1, int32_t cst, int sx, const int32_t *x, int sy, const int32_t *y, int mul2, int32_t *out)
{
    /* lwe.jl:63-82; XOR/XNOR: (x + y) * 2 is a wrapping multiply of every word (lwe.jl:77-80) */
    for (int w = 0; w < n1; w++) {
        int32_t s;
        if (mul2) {
            s = wmul(wadd(x[w], y[w]), 2);
            s = sx > 0 ? s : wneg(s);
        } else {
            s = wadd(sx > 0 ? x[w] : wneg(x[w]), sy > 0 ? y[w] : wneg(y[w]));
        }
        out[w] = s;
    }
    out[n1 - 1] = wadd(out[n1 - 1], cst);
}

typedef struct {
    orc_params P;
    const double *bk_re, *bk_im;
    const int32_t *bk_i32;
    const int32_t *ks;
    int32_t mode;
} orc_keys;

/* One gate, gates.jl:15-177.  x,y,z,out: [n+1]. */
int orc_gate(const orc_params *P, const double *bk_re, const double *bk_im, const int32_t *bk_i32,
             const int32_t *ks, int32_t mode, int32_t op, const int32_t *x, const int32_t *y,
             const int32_t *z, int32_t *out, double *margin)
{
    const int n = P->n, n1 = n + 1, kN = P->k * P->N;
    const int32_t mu = orc_encode_message(1, 3);                            /* gates.jl:17 */
    int32_t tmp[8192 + 1], u1[ORC_MAX_K * ORC_MAX_N + 1], u2[ORC_MAX_K * ORC_MAX_N + 1];
    if (n1 > 8193) return -1;
    double m1 = 0, m2 = 0;
    if (margin) *margin = 0;
    if (op == ORC_NOT) {                                                    /* gates.jl:76-79 */
        for (int w = 0; w < n1; w++) out[w] = wneg(x[w]);
        return 0;
    }
    if (op == ORC_COPY) { memcpy(out, x, sizeof(int32_t) * (size_t)n1); return 0; }
    if (op == ORC_CONST0 || op == ORC_CONST1) {                             /* gates.jl:91-93 */
        memset(out, 0, sizeof(int32_t) * (size_t)n);
        out[n] = orc_encode_message(op == ORC_CONST1 ? 1 : -1, 3);
        return 0;
    }
    if (op == ORC_MUX) {                                                    /* gates.jl:163-177 */
        affine(n1, orc_encode_message(-1, 3), 1, x, 1, y, 0, tmp);          /* :166 */
        if (orc_bootstrap_wo_keyswitch(P, bk_re, bk_im, bk_i32, mode, mu, tmp, u1, margin ? &m1 : NULL)) return -1;
        affine(n1, orc_encode_message(-1, 3), -1, x, 1, z, 0, tmp);         /* :170 */
        if (orc_bootstrap_wo_keyswitch(P, bk_re, bk_im, bk_i32, mode, mu, tmp, u2, margin ? &m2 : NULL)) return -1;
        for (int w = 0; w <= kN; w++) u1[w] = wadd(u1[w], u2[w]);           /* :174 */
        u1[kN] = wadd(u1[kN], orc_encode_message(1, 3));
        orc_keyswitch(P, ks, u1, out);                                      /* :176 */
        if (margin) *margin = m1 > m2 ? m1 : m2;
        return 0;
    }
    int cmu, clog, sx, sy, mul2;
    if (gate_prologue(op, &cmu, &clog, &sx, &sy, &mul2)) return -1;
    affine(n1, orc_encode_message(cmu, clog), sx, x, sy, y, mul2, tmp);
    if (orc_bootstrap_wo_keyswitch(P, bk_re, bk_im, bk_i32, mode, mu, tmp, u1, margin)) return -1;  /* bootstrap.jl:92-95 */
    orc_keyswitch(P, ks, u1, out);
    return 0;
}

/* Batch driver: B independent gates, optionally multi-threaded (one gate per thread).
 * in0/in1/in2/out: [B][n+1]; returns the worst rounding margin seen (mode 0) in *margin. */
int orc_gates_batch(const orc_params *P, const double *bk_re, const double *bk_im, const int32_t *bk_i32,
                    const int32_t *ks, int32_t mode, const uint8_t *ops, const int32_t *in0,
                    const int32_t *in1, const int32_t *in2, int32_t *out, int64_t B, int32_t nthreads,
                    double *margin)
{
    const size_t n1 = (size_t)P->n + 1;
    int rc = 0;
    double worst = 0;
    if (mode == 0 && !get_plan(P->N)) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(max : worst)
    for (int64_t g = 0; g < B; g++) {
        double m = 0;
        int r = orc_gate(P, bk_re, bk_im, bk_i32, ks, mode, ops[g], in0 + g * n1,
                         in1 ? in1 + g * n1 : NULL, in2 ? in2 + g * n1 : NULL, out + g * n1, &m);
        if (r) rc = r;
        if (m > worst) worst = m;
    }
    if (margin) *margin = worst;
    return rc;
}

/* Batch of bootstrap(bk, ks, mu, x) / bootstrap_wo_keyswitch — bootstrap.jl:69-95.
 * out: [B][n+1] if with_ks else [B][kN+1]. */
int orc_bootstrap_batch(const orc_params *P, const double *bk_re, const double *bk_im, const int32_t *bk_i32,
                        const int32_t *ks, int32_t mode, int32_t mu, const int32_t *in, int32_t *out,
                        int64_t B, int32_t with_ks, int32_t nthreads)
{
    const size_t n1 = (size_t)P->n + 1, e1 = (size_t)P->k * P->N + 1;
    int rc = 0;
    if (mode == 0 && !get_plan(P->N)) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t g = 0; g < B; g++) {
        int32_t u[ORC_MAX_K * ORC_MAX_N + 1];
        if (orc_bootstrap_wo_keyswitch(P, bk_re, bk_im, bk_i32, mode, mu, in + g * n1, u, NULL)) rc = -1;
        if (with_ks) orc_keyswitch(P, ks, u, out + g * n1);
        else memcpy(out + g * e1, u, sizeof(int32_t) * e1);
    }
    return rc;
}

int orc_keyswitch_batch(const orc_params *P, const int32_t *ks, const int32_t *in, int32_t *out, int64_t B)
{
    const size_t n1 = (size_t)P->n + 1, e1 = (size_t)P->k * P->N + 1;
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < B; g++) orc_keyswitch(P, ks, in + g * e1, out + g * n1);
    return 0;
}

/* =============================================================================================
 * Multi-key path (config 5): mk_internals.jl:348-411, 464-515 ; mk_gates.jl:7-12.
 * MK samples are flattened as [parties][n] masks (party-major: a[:, p] contiguous) then b.
 * MK bootstrap key per (party i, bit j): x[l][P], y[l][P], c0[l], c1[l] polys, flattened as
 *   [P (party i)][n (bit j)][ x: l*P | y: l*P | c0: l | c1: l ][N]   (x[p_idx][q] at p_idx*P+q)
 * (mode 0 consumes spectra of exactly that layout with N/2 complex values per poly.)
 * mask_size k = 1 is hard-wired in the reference's MK code (mk_internals.jl:89-91,129-131).
 * ===========================================================================================*/
static inline size_t mk_polys_per_key(int l, int Pn) { return (size_t)2 * l * Pn + 2 * l; }

/* per-thread scratch blocks of the multi-key functions, grown to what (parties, l, N) need and kept for the thread's life */
static void *mk_ws(int slot, size_t bytes)
{
    static _Thread_local void *blk[6];
    static _Thread_local size_t cap[6];
    if (bytes > cap[slot]) {
        free(blk[slot]);
        blk[slot] = malloc(bytes);
        cap[slot] = blk[slot] ? bytes : 0;
    }
    return blk[slot];
}

/* mk_internals.jl:348-391 — result added into acc (mk_mux_rotate :469) */
static void mk_extern_mul_add(const orc_params *P, int Pn, int party, const int32_t *temp /*[(Pn+1)][N]*/,
                              const double *kre, const double *kim, const int32_t *ki32, int mode,
                              int32_t *acc, double *margin)
{
    const int N = P->N, M = N / 2, l = P->l;
    /* decompose all Pn masks and b: dec[(i)][p][N], i = 0..Pn (Pn == b)      :355-356 */
    int32_t *dec = mk_ws(0, sizeof(int32_t) * (size_t)(Pn + 1) * l * N);
    for (int i = 0; i <= Pn; i++) orc_decompose(temp + (size_t)i * N, N, l, P->log2Bg, dec + (size_t)i * l * N);
    const size_t X0 = 0, Y0 = (size_t)l * Pn, C0 = (size_t)2 * l * Pn, C1 = C0 + l;

    if (mode == 1) {
        uint32_t *out = mk_ws(1, sizeof(uint32_t) * (size_t)(Pn + 1) * N);
        memset(out, 0, sizeof(uint32_t) * (size_t)(Pn + 1) * N);
        for (int i = 0; i < Pn; i++) {
            for (int p = 0; p < l; p++) {
                const int32_t *d = dec + ((size_t)i * l + p) * N;
                if (i != party)     /* a'_i = sum_p da[p,i] * y[p,party]                    :377-378 */
                    negacyclic_mac_exact(d, ki32 + (Y0 + (size_t)p * Pn + party) * N, N, out + (size_t)i * N);
                /* a'_party += da[p,i] * y[p,i]                                             :373-374 */
                negacyclic_mac_exact(d, ki32 + (Y0 + (size_t)p * Pn + i) * N, N, out + (size_t)party * N);
                /* b' += da[p,i] * x[p,i]                                                   :382-383 */
                negacyclic_mac_exact(d, ki32 + (X0 + (size_t)p * Pn + i) * N, N, out + (size_t)Pn * N);
            }
        }
        for (int p = 0; p < l; p++) {
            const int32_t *d = dec + ((size_t)Pn * l + p) * N;
            negacyclic_mac_exact(d, ki32 + (C1 + p) * N, N, out + (size_t)party * N);       /* :375-376 */
            negacyclic_mac_exact(d, ki32 + (C0 + p) * N, N, out + (size_t)Pn * N);          /* :384-385 */
        }
        for (size_t j = 0; j < (size_t)(Pn + 1) * N; j++) acc[j] = (int32_t)((uint32_t)acc[j] + out[j]);
        return;
    }
    /* mode 0: the reference inverse-transforms every product separately and sums in Int32
     * (:359-366 explains why); restated literally. */
    double *dre = mk_ws(2, sizeof(double) * (size_t)(Pn + 1) * l * M);
    double *dim = mk_ws(3, sizeof(double) * (size_t)(Pn + 1) * l * M);
    for (int i = 0; i <= Pn; i++)
        for (int p = 0; p < l; p++)                                                         /* :368-369 */
            orc_forward_transform(dec + ((size_t)i * l + p) * N, N, dre + ((size_t)i * l + p) * M, dim + ((size_t)i * l + p) * M);
    double pr[ORC_MAX_N / 2], pi_[ORC_MAX_N / 2];
    int32_t prod[ORC_MAX_N];
#define MK_PROD_ADD(DI, KOFF, DST)                                                             \
    do {                                                                                       \
        const double *ar = dre + (size_t)(DI) * M, *ai = dim + (size_t)(DI) * M;               \
        const double *br = kre + (size_t)(KOFF) * M, *bi = kim + (size_t)(KOFF) * M;           \
        for (int f = 0; f < M; f++) {                                                          \
            pr[f] = ar[f] * br[f] - ai[f] * bi[f];                                             \
            pi_[f] = ar[f] * bi[f] + ai[f] * br[f];                                            \
        }                                                                                      \
        double mg = 0;                                                                         \
        orc_inverse_transform(pr, pi_, N, prod, margin ? &mg : NULL);                          \
        if (margin && mg > *margin) *margin = mg;                                              \
        int32_t *dst = acc + (size_t)(DST) * N;                                                \
        for (int j = 0; j < N; j++) dst[j] = wadd(dst[j], prod[j]);                            \
    } while (0)
    for (int i = 0; i < Pn; i++) {
        for (int p = 0; p < l; p++) {
            const size_t di = (size_t)i * l + p;
            if (i != party) MK_PROD_ADD(di, Y0 + (size_t)p * Pn + party, i);
            MK_PROD_ADD(di, Y0 + (size_t)p * Pn + i, party);
            MK_PROD_ADD(di, X0 + (size_t)p * Pn + i, Pn);
        }
    }
    for (int p = 0; p < l; p++) {
        const size_t di = (size_t)Pn * l + p;
        MK_PROD_ADD(di, C1 + p, party);
        MK_PROD_ADD(di, C0 + p, Pn);
    }
#undef MK_PROD_ADD
}

/* mk_internals.jl:498-509, 488-495, 473-485, 464-470, 88-95.
 *   x: [Pn][n] masks then b ; out: extracted [Pn][N] masks then b. */
int orc_mk_bootstrap_wo_keyswitch(const orc_params *P, int32_t Pn, const double *bk_re, const double *bk_im,
                                  const int32_t *bk_i32, int32_t mode, int32_t mu, const int32_t *x,
                                  int32_t *out, double *margin)
{
    const int N = P->N, n = P->n, l = P->l;
    if (Pn > ORC_MAX_PARTIES || N > ORC_MAX_N || l > ORC_MAX_L_MK) return -1;
    if (mode == 0 && !get_plan(N)) return -1;
    const int log2_2N = ilog2(2 * N);
    int32_t *acc = mk_ws(4, sizeof(int32_t) * (size_t)(Pn + 1) * N), *temp = mk_ws(5, sizeof(int32_t) * (size_t)(Pn + 1) * N);
    int32_t tv[ORC_MAX_N];
    if (!acc || !temp) return -1;
    if (margin) *margin = 0;
    const int32_t barb = orc_decode_message(x[(size_t)Pn * n], log2_2N);      /* :502 */
    for (int j = 0; j < N; j++) tv[j] = mu;                                    /* :506 */
    memset(acc, 0, sizeof(int32_t) * (size_t)(Pn + 1) * N);                    /* :72-79 */
    orc_mul_by_monomial(tv, N, -barb, acc + (size_t)Pn * N);                   /* :491 */
    const size_t ppk = mk_polys_per_key(l, Pn);
    for (int i = 0; i < Pn; i++) {                                             /* :475 party-major */
        for (int j = 0; j < n; j++) {                                          /* :476 */
            const int32_t bara = orc_decode_message(x[(size_t)i * n + j], log2_2N);   /* :503 */
            if (bara == 0) continue;                                           /* :478-480 */
            for (int c = 0; c <= Pn; c++) {                                    /* :468, 82-85 */
                orc_mul_by_monomial(acc + (size_t)c * N, N, bara, temp + (size_t)c * N);
                for (int q = 0; q < N; q++) temp[(size_t)c * N + q] = wsub(temp[(size_t)c * N + q], acc[(size_t)c * N + q]);
            }
            const size_t koff = ((size_t)i * n + j) * ppk;
            mk_extern_mul_add(P, Pn, i, temp, bk_re ? bk_re + koff * (N / 2) : NULL,
                              bk_im ? bk_im + koff * (N / 2) : NULL, bk_i32 ? bk_i32 + koff * N : NULL,
                              mode, acc, margin);                              /* :469 */
        }
    }
    /* mk_tlwe_extract_sample :88-95 — one extracted mask column per party */
    for (int c = 0; c < Pn; c++) {
        const int32_t *p = acc + (size_t)c * N;
        int32_t *o = out + (size_t)c * N;
        o[0] = p[0];
        for (int m = 1; m < N; m++) o[m] = wneg(p[N - m]);
    }
    out[(size_t)Pn * N] = acc[(size_t)Pn * N];
    return 0;
}

/* mk_internals.jl:397-411 — per-party single-key keyswitch with b = 0, b's summed onto sample.b.
 *   ks: [Pn] keys each [N][t][base-1][n+1] ; in: [Pn][N] then b ; out: [Pn][n] then b */
void orc_mk_keyswitch(const orc_params *P, int32_t Pn, const int32_t *ks, const int32_t *in, int32_t *out)
{
    const int N = P->N, n = P->n;
    const size_t ks_words = (size_t)N * P->t * ((1u << P->log2ks) - 1) * (size_t)(n + 1);
    orc_params P1 = *P; P1.k = 1;
    int32_t b = in[(size_t)Pn * N];                                            /* :406 */
    int32_t tin[ORC_MAX_N + 1], tout[8193];
    for (int p = 0; p < Pn; p++) {
        memcpy(tin, in + (size_t)p * N, sizeof(int32_t) * (size_t)N);
        tin[N] = 0;                                                            /* :400 b = 0 */
        orc_keyswitch(&P1, ks + (size_t)p * ks_words, tin, tout);
        memcpy(out + (size_t)p * n, tout, sizeof(int32_t) * (size_t)n);
        b = wadd(b, tout[n]);                                                  /* :409 */
    }
    out[(size_t)Pn * n] = b;
}

/* mk_gates.jl:7-12 + mk_internals.jl:512-515 */
int orc_mk_gate_nand(const orc_params *P, int32_t Pn, const double *bk_re, const double *bk_im,
                     const int32_t *bk_i32, const int32_t *ks, int32_t mode, const int32_t *x,
                     const int32_t *y, int32_t *out, double *margin)
{
    const size_t words = (size_t)Pn * P->n + 1;
    int32_t *tmp = malloc(sizeof(int32_t) * words);
    int32_t *u = malloc(sizeof(int32_t) * ((size_t)Pn * P->N + 1));
    if (!tmp || !u) { free(tmp); free(u); return -1; }
    for (size_t w = 0; w < words; w++) tmp[w] = wsub(wneg(x[w]), y[w]);        /* mk_gates.jl:8-10 */
    tmp[words - 1] = wadd(tmp[words - 1], orc_encode_message(1, 3));
    int rc = orc_mk_bootstrap_wo_keyswitch(P, Pn, bk_re, bk_im, bk_i32, mode, orc_encode_message(1, 3), tmp, u, margin);
    if (!rc) orc_mk_keyswitch(P, Pn, ks, u, out);
    free(tmp); free(u);
    return rc;
}

int orc_mk_gate_nand_batch(const orc_params *P, int32_t Pn, const double *bk_re, const double *bk_im,
                           const int32_t *bk_i32, const int32_t *ks, int32_t mode, const int32_t *in0,
                           const int32_t *in1, int32_t *out, int64_t B, int32_t nthreads, double *margin)
{
    const size_t words = (size_t)Pn * P->n + 1;
    int rc = 0;
    double worst = 0;
    if (mode == 0 && !get_plan(P->N)) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(max : worst)
    for (int64_t g = 0; g < B; g++) {
        double m = 0;
        if (orc_mk_gate_nand(P, Pn, bk_re, bk_im, bk_i32, ks, mode, in0 + g * words, in1 + g * words, out + g * words, &m)) rc = -1;
        if (m > worst) worst = m;
    }
    if (margin) *margin = worst;
    return rc;
}

int32_t orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
