# TFHEMI355X.jl — ccall shim that puts libtfhe_mi355x.so behind TFHE.jl's own gate API.
#
# NOT EXECUTED IN THE BUILD IMAGE (no Julia there): kept literal and small so it can be reviewed by
# reading against include/tfhe_mi355x.h.  Usage (on a box with Julia, TFHE.jl and an MI355X):
#
#     using TFHE, Random
#     include("julia/TFHEMI355X.jl"); using .TFHEMI355X
#     rng = MersenneTwister(123)
#     secret_key, cloud_key = make_key_pair(rng)
#     gck = GpuCloudKey(cloud_key)                      # flattens + uploads the keys once
#     r = gate_nand(gck, encrypt(rng, secret_key, true), encrypt(rng, secret_key, false))
#     rs = gate_nand(gck, xs, ys)                       # Vector{LweSample}: ONE batched GPU call
#
# Every method has the name and argument order of the TFHE.jl function it replaces
# (src/gates.jl:15-177); the cloud-key argument is a GpuCloudKey instead of a CloudKey.
module TFHEMI355X

using TFHE
using TFHE: LweSample, LweParams, CloudKey, SchemeParameters

export GpuCloudKey, gate_nand, gate_or, gate_and, gate_xor, gate_xnor, gate_not, gate_constant,
       gate_nor, gate_andny, gate_andyn, gate_orny, gate_oryn, gate_mux, gates_batch

const LIB = get(ENV, "TFHE_MI355X_LIB", joinpath(@__DIR__, "..", "tfhe.jl_amd", "lib", "libtfhe_mi355x.so"))

# include/tfhe_mi355x.h: struct tfhe_params
struct TfheParams
    n::Int32; N::Int32; k::Int32; bs_l::Int32; bs_log2_base::Int32
    ks_t::Int32; ks_log2_base::Int32; parties::Int32
end

# opcodes (include/tfhe_mi355x.h: TFHE_GATE_*)
const NAND, OR, AND, XOR, XNOR, NOT, NOR, ANDNY, ANDYN, ORNY, ORYN, MUX, CONST0, CONST1, COPY =
    UInt8.(0:14)

function check(ctx::Ptr{Cvoid}, rc::Int32)
    rc == 0 && return
    msg = unsafe_string(ccall((:tfhe_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
    error("tfhe_mi355x error $rc: $msg")
end

mutable struct GpuCloudKey
    params::SchemeParameters
    ctx::Ptr{Cvoid}

    function GpuCloudKey(ck::CloudKey; device::Integer=0)
        p = ck.params
        tp = TfheParams(p.lwe_size, p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length,
                        p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base, p.max_parties)
        ctxref = Ref{Ptr{Cvoid}}(C_NULL)
        rc = ccall((:tfhe_ctx_create, LIB), Int32, (Ref{TfheParams}, Int32, Ref{Ptr{Cvoid}}),
                   tp, Int32(device), ctxref)
        check(Ptr{Cvoid}(C_NULL), rc)
        ctx = ctxref[]

        # BootstrapKey: only the transformed form exists (src/bootstrap.jl:12-14).  Flatten
        # key[i].samples[p, j].a[c].coeffs (Complex{Float64}[N/2]) to [n][l][k+1][k+1][N/2].
        bk = ck.bootstrap_key
        n, l, k1, M = p.lwe_size, p.bs_decomp_length, p.tlwe_mask_size + 1, p.tlwe_polynomial_degree ÷ 2
        spectra = Array{Complex{Float64}}(undef, M, k1, k1, l, n)        # column-major: M fastest
        for i in 1:n, pp in 1:l, j in 1:k1, c in 1:k1
            spectra[:, c, j, pp, i] .= bk.key[i].samples[pp, j].a[c].coeffs
        end
        GC.@preserve spectra check(ctx, ccall((:tfhe_load_bootstrap_key_c128, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Complex{Float64}}), ctx, spectra))

        # KeyswitchKey: key[h, j, i]::LweSample (src/keyswitch.jl:12,35-38) -> [kN][t][base-1][n+1]
        ks = ck.keyswitch_key
        base1, t, kN = size(ks.key)
        flat = Array{Int32}(undef, n + 1, base1, t, kN)
        for i in 1:kN, j in 1:t, h in 1:base1
            s = ks.key[h, j, i]
            flat[1:n, h, j, i] .= s.a
            flat[n + 1, h, j, i] = s.b
        end
        GC.@preserve flat check(ctx, ccall((:tfhe_load_keyswitch_key, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Int32}), ctx, flat))

        gck = new(p, ctx)
        finalizer(g -> ccall((:tfhe_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), g.ctx), gck)
        gck
    end
end

# LweSample <-> flat Int32[n+1] (a then b), include/tfhe_mi355x.h
function flatten(xs::AbstractVector{LweSample})
    n = xs[1].params.size
    m = Array{Int32}(undef, n + 1, length(xs))
    for (g, x) in enumerate(xs)
        m[1:n, g] .= x.a
        m[n + 1, g] = x.b
    end
    m
end

unflatten(m::Matrix{Int32}, params::LweParams) =
    # current_variance is write-only bookkeeping in the reference (SURVEY §5); 0.0 as tlwe.jl:58 does
    [LweSample(params, m[1:end-1, g], m[end, g], 0.) for g in 1:size(m, 2)]

"""
    gates_batch(gck, opcodes, xs, ys, zs)

`length(opcodes)` independent gates in one GPU call (tfhe_gates_batch).
"""
function gates_batch(gck::GpuCloudKey, opcodes::Vector{UInt8}, xs, ys=nothing, zs=nothing)
    B = length(opcodes)
    params = LweParams(gck.params.lwe_size)
    fx = xs === nothing ? nothing : flatten(xs)
    fy = ys === nothing ? nothing : flatten(ys)
    fz = zs === nothing ? nothing : flatten(zs)
    out = Array{Int32}(undef, gck.params.lwe_size + 1, B)
    ptr(a) = a === nothing ? Ptr{Int32}(C_NULL) : pointer(a)
    GC.@preserve fx fy fz out opcodes check(gck.ctx, ccall((:tfhe_gates_batch, LIB), Int32,
        (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64),
        gck.ctx, opcodes, ptr(fx), ptr(fy), ptr(fz), out, B))
    unflatten(out, params)
end

# scalar and vector methods with the reference's names (src/gates.jl)
for (name, op) in ((:gate_nand, NAND), (:gate_or, OR), (:gate_and, AND), (:gate_xor, XOR),
                   (:gate_xnor, XNOR), (:gate_nor, NOR), (:gate_andny, ANDNY), (:gate_andyn, ANDYN),
                   (:gate_orny, ORNY), (:gate_oryn, ORYN))
    @eval begin
        $name(gck::GpuCloudKey, x::LweSample, y::LweSample) = gates_batch(gck, [$op], [x], [y])[1]
        $name(gck::GpuCloudKey, xs::AbstractVector{LweSample}, ys::AbstractVector{LweSample}) =
            gates_batch(gck, fill($op, length(xs)), xs, ys)
    end
end

gate_mux(gck::GpuCloudKey, x::LweSample, y::LweSample, z::LweSample) =
    gates_batch(gck, [MUX], [x], [y], [z])[1]
gate_mux(gck::GpuCloudKey, xs::AbstractVector{LweSample}, ys::AbstractVector{LweSample},
         zs::AbstractVector{LweSample}) = gates_batch(gck, fill(MUX, length(xs)), xs, ys, zs)

# not bootstrapped (src/gates.jl:76-93): cheap on the host, no device round trip needed
gate_not(gck::GpuCloudKey, x::LweSample) = TFHE.LweSample(x.params, -x.a, -x.b, x.current_variance)
gate_constant(gck::GpuCloudKey, value::Bool) =
    TFHE.lwe_noiseless_trivial(TFHE.encode_message(value ? 1 : -1, 8), LweParams(gck.params.lwe_size))

# ---- multi-key (src/mk_gates.jl:7-12) ----------------------------------------------------------------------
# GpuMKCloudKey(ck::TFHE.MKCloudKey): flatten ck.bootstrap_key.key[j, i] (x[l, P], y[l, P], c0[l], c1[l] spectra;
# src/mk_internals.jl:274-288,442-461) by applying TFHE.inverse_transform to every spectrum (exact: integers) into
# Int32 [P][n][2lP + 2l][N] (x[p, q] at p*P + q, then y, c0, c1), call tfhe_mk_load_bootstrap_key_i32, flatten the
# P keyswitch keys as in GpuCloudKey and call tfhe_mk_load_keyswitch_key; an MKLweSample is the Int32 column
# [a[:, 1]; a[:, 2]; ...; b] (src/mk_internals.jl:6-18).  mk_gate_nand(gck, x, y) is then one ccall:
#
#   ccall((:tfhe_mk_gate_nand_batch, LIB), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64),
#         gck.ctx, fx, fy, out, B)

end # module
