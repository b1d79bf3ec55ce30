/* abi_client.c — a plain C client of the drop-in boundary (include/tfhe_mi355x.h): no Python, no C++.
 * Loads a cloud key and LWE operands from raw little-endian files written by the test, runs
 * tfhe_gates_batch and writes the result; the pytest side compares it with the oracle.
 *
 *   abi_client <lib.so> <dir>      dir holds params.i32 (8 words), bk.i32, ks.i32, ops.u8, in0.i32, in1.i32, in2.i32
 *                                  -> writes out.i32, prints "ok <B>"
 * Built with:  gcc -O2 -I include tests/c_abi/abi_client.c -ldl -o tests/c_abi/abi_client
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tfhe_mi355x.h"

static void *slurp(const char *dir, const char *name, size_t *bytes)
{
    char path[4096];
    snprintf(path, sizeof path, "%s/%s", dir, name);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *p = malloc((size_t)n ? (size_t)n : 1);
    if (fread(p, 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "short read %s\n", path); exit(2); }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

#define SYM(type, name) type name = (type)dlsym(lib, #name); if (!name) { fprintf(stderr, "missing symbol %s\n", #name); return 3; }

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s <lib.so> <dir>\n", argv[0]); return 2; }
    void *lib = dlopen(argv[1], RTLD_NOW);
    if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    typedef int32_t (*create_t)(const tfhe_params *, int32_t, tfhe_ctx **);
    typedef void (*destroy_t)(tfhe_ctx *);
    typedef const char *(*err_t)(const tfhe_ctx *);
    typedef int32_t (*loadk_t)(tfhe_ctx *, const int32_t *);
    typedef int32_t (*gates_t)(tfhe_ctx *, const uint8_t *, const int32_t *, const int32_t *, const int32_t *, int32_t *, int64_t);
    SYM(create_t, tfhe_ctx_create) SYM(destroy_t, tfhe_ctx_destroy) SYM(err_t, tfhe_last_error)
    SYM(loadk_t, tfhe_load_bootstrap_key_i32) SYM(loadk_t, tfhe_load_keyswitch_key) SYM(gates_t, tfhe_gates_batch)

    size_t nb;
    int32_t *pw = slurp(argv[2], "params.i32", &nb);
    if (nb != 8 * sizeof(int32_t)) { fprintf(stderr, "params.i32 must hold 8 words\n"); return 2; }
    tfhe_params P = {pw[0], pw[1], pw[2], pw[3], pw[4], pw[5], pw[6], pw[7]};
    tfhe_ctx *ctx = NULL;
    if (tfhe_ctx_create(&P, 0, &ctx)) { fprintf(stderr, "ctx_create: %s\n", tfhe_last_error(NULL)); return 4; }
    int32_t *bk = slurp(argv[2], "bk.i32", &nb);
    if (tfhe_load_bootstrap_key_i32(ctx, bk)) { fprintf(stderr, "load bk: %s\n", tfhe_last_error(ctx)); return 4; }
    int32_t *ks = slurp(argv[2], "ks.i32", &nb);
    if (tfhe_load_keyswitch_key(ctx, ks)) { fprintf(stderr, "load ks: %s\n", tfhe_last_error(ctx)); return 4; }
    size_t nops;
    uint8_t *ops = slurp(argv[2], "ops.u8", &nops);
    int32_t *in0 = slurp(argv[2], "in0.i32", &nb), *in1 = slurp(argv[2], "in1.i32", &nb), *in2 = slurp(argv[2], "in2.i32", &nb);
    if (nb != nops * (size_t)(P.n + 1) * 4) { fprintf(stderr, "operand size mismatch\n"); return 2; }
    int32_t *out = malloc(nb);
    if (tfhe_gates_batch(ctx, ops, in0, in1, in2, out, (int64_t)nops)) { fprintf(stderr, "gates_batch: %s\n", tfhe_last_error(ctx)); return 5; }
    /* misuse must be reported, not crash: a bad opcode */
    uint8_t bad = 200;
    if (tfhe_gates_batch(ctx, &bad, in0, in1, in2, out, 1) != TFHE_ERR_INVALID_ARG) { fprintf(stderr, "bad opcode not rejected\n"); return 6; }
    if (tfhe_gates_batch(ctx, ops, in0, in1, in2, out, (int64_t)nops)) return 5;
    char path[4096];
    snprintf(path, sizeof path, "%s/out.i32", argv[2]);
    FILE *f = fopen(path, "wb");
    fwrite(out, 1, nb, f);
    fclose(f);
    tfhe_ctx_destroy(ctx);
    printf("ok %zu\n", nops);
    return 0;
}
