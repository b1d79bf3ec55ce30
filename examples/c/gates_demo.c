/* gates_demo.c — the whole flow through the C ABI alone (no Python, no oracle): draw secret bits, let the GPU generate
 * the cloud key (tfhe_keygen_cloud_key), encrypt bits on the host (lwe.jl:49-55), evaluate every two-input gate, NOT and
 * MUX on all input combinations in one batch (tfhe_gates_batch), decrypt (api.jl:167-169) and compare with the truth
 * tables of gates.jl.  What a C (or, through ccall, Julia) caller of libtfhe_mi355x.so does.
 *
 *   gcc -O2 -I include examples/c/gates_demo.c -ldl -lm -o gates_demo && ./gates_demo tfhe.jl_amd/lib/libtfhe_mi355x.so
 *
 * Exit status: 0 = every gate decrypted to its truth value, 3 = a wrong bit, 4 = no HIP device, other = ABI error. */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include "tfhe_mi355x.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t next_u64(void)     /* splitmix64 */
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double next_gaussian(void)
{
    const double u1 = ((double)(next_u64() >> 11) + 0.5) / 9007199254740992.0, u2 = ((double)(next_u64() >> 11) + 0.5) / 9007199254740992.0;
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

#define LOAD(name) name##_t p_##name = (name##_t)dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
typedef int32_t (*tfhe_device_count_t)(void);
typedef int32_t (*tfhe_ctx_create_t)(const tfhe_params *, int32_t, tfhe_ctx **);
typedef void (*tfhe_ctx_destroy_t)(tfhe_ctx *);
typedef const char *(*tfhe_last_error_t)(const tfhe_ctx *);
typedef int32_t (*tfhe_keygen_cloud_key_t)(tfhe_ctx *, const int32_t *, const int32_t *, double, double, const uint32_t *, int32_t *, int32_t *);
typedef int32_t (*tfhe_gates_batch_t)(tfhe_ctx *, const uint8_t *, const int32_t *, const int32_t *, const int32_t *, int32_t *, int64_t);
typedef int32_t (*tfhe_gates_batch_submit_t)(tfhe_ctx *, const uint8_t *, const int32_t *, const int32_t *, const int32_t *, int32_t *, int64_t, int32_t *);
typedef int32_t (*tfhe_gates_batch_wait_t)(tfhe_ctx *, int32_t);
typedef int32_t (*tfhe_host_alloc_t)(size_t, void **);
typedef void (*tfhe_host_free_t)(void *);

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s <path to libtfhe_mi355x.so>\n", argv[0]); return 2; }
    void *lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    LOAD(tfhe_device_count) LOAD(tfhe_ctx_create) LOAD(tfhe_ctx_destroy) LOAD(tfhe_last_error) LOAD(tfhe_keygen_cloud_key) LOAD(tfhe_gates_batch)
    LOAD(tfhe_gates_batch_submit) LOAD(tfhe_gates_batch_wait) LOAD(tfhe_host_alloc) LOAD(tfhe_host_free)
    if (p_tfhe_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 4; }

    /* tfhe_parameters_80 (api.jl:30-45) */
    const int n = 500, N = 1024, k = 1;
    const double lwe_noise = pow(2.0, -15) * sqrt(2.0 / 3.14159265358979323846), bs_noise = 9.0e-9 * sqrt(2.0 / 3.14159265358979323846);
    tfhe_params P = {n, N, k, 2, 10, 8, 2, 1};
    tfhe_ctx *ctx = NULL;
    if (p_tfhe_ctx_create(&P, 0, &ctx)) { fprintf(stderr, "ctx_create: %s\n", p_tfhe_last_error(NULL)); return 5; }

    int32_t *lwe_key = malloc(sizeof(int32_t) * n), *tlwe_key = malloc(sizeof(int32_t) * k * N);
    for (int i = 0; i < n; i++) lwe_key[i] = (int32_t)(next_u64() & 1);
    for (int i = 0; i < k * N; i++) tlwe_key[i] = (int32_t)(next_u64() & 1);
    /* six seed words: two for the (public) masks, four — as secret as the key itself — for the noise; a real client draws
     * them from a cryptographic source, this demo from its test generator */
    uint32_t seed[6];
    for (int i = 0; i < 6; i++) seed[i] = (uint32_t)next_u64();
    if (p_tfhe_keygen_cloud_key(ctx, lwe_key, tlwe_key, bs_noise, lwe_noise, seed, NULL, NULL)) {
        fprintf(stderr, "keygen: %s\n", p_tfhe_last_error(ctx));
        return 6;
    }

    /* 13 gate kinds x 8 input combinations */
    enum { KINDS = 14, B = KINDS * 8 };
    const int kinds[KINDS] = {TFHE_GATE_NAND, TFHE_GATE_OR, TFHE_GATE_AND, TFHE_GATE_XOR, TFHE_GATE_XNOR, TFHE_GATE_NOT, TFHE_GATE_NOR,
                              TFHE_GATE_ANDNY, TFHE_GATE_ANDYN, TFHE_GATE_ORNY, TFHE_GATE_ORYN, TFHE_GATE_MUX, TFHE_GATE_CONST0, TFHE_GATE_CONST1};
    uint8_t ops[B];
    int bits[3][B];
    int32_t *in[3], *out = malloc(sizeof(int32_t) * B * (n + 1));
    for (int o = 0; o < 3; o++) {
        in[o] = malloc(sizeof(int32_t) * B * (n + 1));
        for (int g = 0; g < B; g++) {
            const int bit = bits[o][g] = ((g & 7) >> o) & 1;
            int32_t *s = in[o] + (size_t)g * (n + 1);
            uint32_t b = bit ? (1u << 29) : 0u - (1u << 29);                      /* encode_message(+-1, 8) */
            b += (uint32_t)(int32_t)trunc(next_gaussian() * lwe_noise * 4294967296.0);
            for (int i = 0; i < n; i++) {
                s[i] = (int32_t)(uint32_t)next_u64();
                if (lwe_key[i]) b += (uint32_t)s[i];
            }
            s[n] = (int32_t)b;
        }
    }
    for (int g = 0; g < B; g++) ops[g] = (uint8_t)kinds[g / 8];
    if (p_tfhe_gates_batch(ctx, ops, in[0], in[1], in[2], out, B)) { fprintf(stderr, "gates_batch: %s\n", p_tfhe_last_error(ctx)); return 7; }

    /* the streaming form: the same batch submitted three times, two in flight, results in page-locked buffers */
    {
        int32_t *sout[3] = {NULL, NULL, NULL};
        int32_t ticket[3];
        const size_t bytes = sizeof(int32_t) * B * (n + 1);
        for (int k = 0; k < 3; k++)
            if (p_tfhe_host_alloc(bytes, (void **)&sout[k])) { fprintf(stderr, "host_alloc failed\n"); return 8; }
        for (int k = 0; k < 3; k++)
            if (p_tfhe_gates_batch_submit(ctx, ops, in[0], in[1], in[2], sout[k], B, &ticket[k])) { fprintf(stderr, "submit: %s\n", p_tfhe_last_error(ctx)); return 8; }
        for (int k = 0; k < 3; k++)
            if (p_tfhe_gates_batch_wait(ctx, ticket[k])) { fprintf(stderr, "wait: %s\n", p_tfhe_last_error(ctx)); return 8; }
        for (int k = 0; k < 3; k++) {
            if (memcmp(sout[k], out, bytes)) { fprintf(stderr, "streamed batch %d differs from the blocking call\n", k); return 9; }
            p_tfhe_host_free(sout[k]);
        }
    }

    int wrong = 0;
    for (int g = 0; g < B; g++) {
        const int x = bits[0][g], y = bits[1][g], z = bits[2][g];
        int want;
        switch (kinds[g / 8]) {
        case TFHE_GATE_NAND: want = !(x && y); break;
        case TFHE_GATE_OR: want = x || y; break;
        case TFHE_GATE_AND: want = x && y; break;
        case TFHE_GATE_XOR: want = x ^ y; break;
        case TFHE_GATE_XNOR: want = !(x ^ y); break;
        case TFHE_GATE_NOT: want = !x; break;
        case TFHE_GATE_NOR: want = !(x || y); break;
        case TFHE_GATE_ANDNY: want = !x && y; break;
        case TFHE_GATE_ANDYN: want = x && !y; break;
        case TFHE_GATE_ORNY: want = !x || y; break;
        case TFHE_GATE_ORYN: want = x || !y; break;
        case TFHE_GATE_MUX: want = x ? y : z; break;
        case TFHE_GATE_CONST0: want = 0; break;
        default: want = 1; break;
        }
        const int32_t *s = out + (size_t)g * (n + 1);
        uint32_t phase = (uint32_t)s[n];
        for (int i = 0; i < n; i++) if (lwe_key[i]) phase -= (uint32_t)s[i];
        if (((int32_t)phase > 0) != want) wrong++;
    }
    p_tfhe_ctx_destroy(ctx);
    if (wrong) { fprintf(stderr, "%d of %d gates decrypted to the wrong bit\n", wrong, (int)B); return 3; }
    printf("ok: %d gates (%d kinds x 8 input combinations) on a device-generated key\n", (int)B, (int)KINDS);
    return 0;
}
