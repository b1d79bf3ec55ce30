// kernels_gates.hpp — gate prologue / modulus switch / non-bootstrapped gates (gates.jl, bootstrap.jl:74-75).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/tfhe_mi355x.h"
#include "br_core.hpp"

using namespace tfhe;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------

// Per-opcode affine prologue  t = (0, cst) + sx*x + sy*y  [* 2 for XOR/XNOR]   (gates.jl)
struct GateForm {
    int32_t cst;   // constant added to b
    int8_t sx, sy; // +-1 coefficients (after the optional doubling)
    int8_t mul2;   // (x + y) * 2 form (gates.jl:52,64)
    int8_t use_z;  // second operand comes from in2 (MUX second half)
};

__host__ __device__ inline GateForm gate_form(int kind)
{
    // kind: opcode for plain gates; 100 = MUX first half (AND(x,y)), 101 = MUX second half (ANDNY(x,z))
    const int32_t p8 = (int32_t)(1u << 29), p4 = (int32_t)(1u << 30);
    switch (kind) {
    case TFHE_GATE_NAND:  return {p8, -1, -1, 0, 0};
    case TFHE_GATE_OR:    return {p8, 1, 1, 0, 0};
    case TFHE_GATE_AND:   return {-p8, 1, 1, 0, 0};
    case TFHE_GATE_XOR:   return {p4, 1, 1, 1, 0};
    case TFHE_GATE_XNOR:  return {-p4, -1, -1, 1, 0};
    case TFHE_GATE_NOR:   return {-p8, -1, -1, 0, 0};
    case TFHE_GATE_ANDNY: return {-p8, -1, 1, 0, 0};
    case TFHE_GATE_ANDYN: return {-p8, 1, -1, 0, 0};
    case TFHE_GATE_ORNY:  return {p8, -1, 1, 0, 0};
    case TFHE_GATE_ORYN:  return {p8, 1, -1, 0, 0};
    case 100:             return {-p8, 1, 1, 0, 0};   // gates.jl:166
    case 101:             return {-p8, -1, 1, 0, 1};  // gates.jl:170
    default:              return {0, 0, 0, 0, 0};
    }
}

// rot_a[w] / rot_b[w] = rows of the two operands of rotation w (batch mode: the gate index; level mode: wire
// indices), rot_kind[w] = kind (see gate_form); writes bara[w][0..n] (barb last).
__global__ void prologue_kernel(const int32_t *in0, const int32_t *in1, const int32_t *in2,
                                const int32_t *__restrict__ rot_a, const int32_t *__restrict__ rot_b,
                                const uint8_t *__restrict__ rot_kind, int32_t *__restrict__ bara, int n,
                                int log2_2N)
{
    const int w = blockIdx.x;
    const GateForm f = gate_form(rot_kind[w]);
    const int32_t *x = in0 + (size_t)rot_a[w] * (n + 1);
    const int32_t *y = (f.use_z ? in2 : in1) + (size_t)rot_b[w] * (n + 1);
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        uint32_t v;
        if (f.mul2) {
            v = ((uint32_t)x[i] + (uint32_t)y[i]) * 2u;
            if (f.sx < 0) v = 0u - v;
        } else {
            const uint32_t xv = f.sx > 0 ? (uint32_t)x[i] : 0u - (uint32_t)x[i];
            const uint32_t yv = f.sy > 0 ? (uint32_t)y[i] : 0u - (uint32_t)y[i];
            v = xv + yv;
        }
        if (i == n) v += (uint32_t)f.cst;
        // decode_message(v, 2N): numeric-functions.jl:31-34
        const int32_t r = (int32_t)(v + (1u << (32 - log2_2N - 1))) >> (32 - log2_2N);
        bara[(size_t)w * (n + 1) + i] = r;
    }
}

// modulus switch only (tfhe_bootstrap_batch): bara[w][i] = decode_message(in[w][i], 2N)
__global__ void modswitch_kernel(const int32_t *__restrict__ in, int32_t *__restrict__ bara, int n, int log2_2N)
{
    const size_t w = blockIdx.x;
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        const uint32_t v = (uint32_t)in[w * (n + 1) + i];
        bara[w * (n + 1) + i] = (int32_t)(v + (1u << (32 - log2_2N - 1))) >> (32 - log2_2N);
    }
}

// gate_not / gate_constant / copy (gates.jl:76-93)
__global__ void trivial_gates_kernel(const int32_t *in0, const int32_t *__restrict__ src_rows,
                                     const int32_t *__restrict__ dst_rows, const uint8_t *__restrict__ ops,
                                     int32_t *out, int n)
{
    const size_t gs = (size_t)src_rows[blockIdx.x], gd = (size_t)dst_rows[blockIdx.x];
    const int op = ops[blockIdx.x];
    for (int i = threadIdx.x; i <= n; i += blockDim.x) {
        uint32_t v;
        if (op == TFHE_GATE_NOT) v = 0u - (uint32_t)in0[gs * (n + 1) + i];
        else if (op == TFHE_GATE_COPY) v = (uint32_t)in0[gs * (n + 1) + i];
        else v = (i == n) ? (op == TFHE_GATE_CONST1 ? (1u << 29) : 0u - (1u << 29)) : 0u;
        out[gd * (n + 1) + i] = (int32_t)v;
    }
}

