# TFHEMI355X.jl — ccall shim that puts libtfhe_mi355x.so behind TFHE.jl's own gate API.
#
# NOT EXECUTED IN THE BUILD IMAGE (no Julia there): kept literal and small so it can be reviewed by
# reading against include/tfhe_mi355x.h.  Usage (on a box with Julia, TFHE.jl and one or more MI355X):
#
#     using TFHE, Random
#     include("julia/TFHEMI355X.jl"); using .TFHEMI355X
#     rng = MersenneTwister(123)
#     secret_key, cloud_key = make_key_pair(rng)
#     gck = GpuCloudKey(cloud_key)                      # flattens + uploads the keys once (device 0)
#     gck8 = GpuCloudKey(cloud_key; devices=0:7)        # keys replicated on 8 GPUs, every batch call split over them
#     r = gate_nand(gck, encrypt(rng, secret_key, true), encrypt(rng, secret_key, false))
#     rs = gate_nand(gck8, xs, ys)                      # Vector{LweSample}: ONE batched call (the analogue of
#                                                       # gate_nand.(cloud_key, xs, ys), docs/src/manual.md:28-35)
#     mck = GpuMKCloudKey(mk_cloud_key)                 # multi-key: MKCloudKey -> device
#     out = mk_gate_nand(mck, x, y)                     # MKLweSample (or vectors of them)
#
# Every method has the name and argument order of the TFHE.jl function it replaces
# (src/gates.jl:15-177, src/mk_gates.jl:7-12); the cloud-key argument is a GpuCloudKey / GpuMKCloudKey.
module TFHEMI355X

using TFHE
using TFHE: LweSample, LweParams, CloudKey, SecretKey, SchemeParameters, MKCloudKey, MKLweSample
using Random: AbstractRNG

export GpuCloudKey, GpuMKCloudKey, gate_nand, gate_or, gate_and, gate_xor, gate_xnor, gate_not, gate_constant,
       gate_nor, gate_andny, gate_andyn, gate_orny, gate_oryn, gate_mux, gates_batch, mk_gate_nand

const LIB = get(ENV, "TFHE_MI355X_LIB", joinpath(@__DIR__, "..", "tfhe.jl_amd", "lib", "libtfhe_mi355x.so"))

# include/tfhe_mi355x.h: struct tfhe_params
struct TfheParams
    n::Int32; N::Int32; k::Int32; bs_l::Int32; bs_log2_base::Int32
    ks_t::Int32; ks_log2_base::Int32; parties::Int32
end

TfheParams(p::SchemeParameters) = TfheParams(
    p.lwe_size, p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length,
    p.bs_log2_base, p.ks_decomp_length, p.ks_log2_base, p.max_parties)

# opcodes (include/tfhe_mi355x.h: TFHE_GATE_*)
const NAND, OR, AND, XOR, XNOR, NOT, NOR, ANDNY, ANDYN, ORNY, ORYN, MUX, CONST0, CONST1, COPY =
    UInt8.(0:14)

function check(ctx::Ptr{Cvoid}, rc::Int32)
    rc == 0 && return
    msg = unsafe_string(ccall((:tfhe_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
    error("tfhe_mi355x error $rc: $msg")
end

# tfhe_ctx_create (one device) or tfhe_ctx_create_multi (keys replicated, batch calls fanned out inside the library)
function create_context(p::SchemeParameters, devices)
    tp = TfheParams(p)
    ctxref = Ref{Ptr{Cvoid}}(C_NULL)
    ids = Int32.(collect(devices))
    rc = if length(ids) == 1
        ccall((:tfhe_ctx_create, LIB), Int32, (Ref{TfheParams}, Int32, Ref{Ptr{Cvoid}}), tp, ids[1], ctxref)
    else
        ccall((:tfhe_ctx_create_multi, LIB), Int32, (Ref{TfheParams}, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
              tp, ids, Int32(length(ids)), ctxref)
    end
    check(Ptr{Cvoid}(C_NULL), rc)
    ctxref[]
end

# KeyswitchKey: key[h, j, i]::LweSample (src/keyswitch.jl:12,35-38) -> Int32 [kN][t][base-1][n+1] (C order)
function flatten_keyswitch_key(ks, n)
    base1, t, kN = size(ks.key)
    flat = Array{Int32}(undef, n + 1, base1, t, kN)                   # column-major: n+1 fastest
    for i in 1:kN, j in 1:t, h in 1:base1
        s = ks.key[h, j, i]
        flat[1:n, h, j, i] .= s.a
        flat[n + 1, h, j, i] = s.b
    end
    flat
end

# BootstrapKey: only the transformed form exists (src/bootstrap.jl:12-14).  Flatten
# key[i].samples[p, j].a[c].coeffs (Complex{Float64}[N/2]) to C order [n][l][k+1][k+1][N/2].
function flatten_bootstrap_spectra(bk, p::SchemeParameters)
    n, l, k1, M = p.lwe_size, p.bs_decomp_length, p.tlwe_mask_size + 1, p.tlwe_polynomial_degree ÷ 2
    spectra = Array{Complex{Float64}}(undef, M, k1, k1, l, n)        # column-major: M fastest
    for i in 1:n, pp in 1:l, j in 1:k1, c in 1:k1
        spectra[:, c, j, pp, i] .= bk.key[i].samples[pp, j].a[c].coeffs
    end
    spectra
end

mutable struct GpuCloudKey
    params::SchemeParameters
    ctx::Ptr{Cvoid}

    function GpuCloudKey(ck::CloudKey; device::Integer=0, devices=nothing)
        p = ck.params
        ctx = create_context(p, devices === nothing ? [device] : devices)
        spectra = flatten_bootstrap_spectra(ck.bootstrap_key, p)
        GC.@preserve spectra check(ctx, ccall((:tfhe_load_bootstrap_key_c128, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Complex{Float64}}), ctx, spectra))
        flat = flatten_keyswitch_key(ck.keyswitch_key, p.lwe_size)
        GC.@preserve flat check(ctx, ccall((:tfhe_load_keyswitch_key, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Int32}), ctx, flat))
        gck = new(p, ctx)
        finalizer(g -> ccall((:tfhe_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), g.ctx), gck)
        gck
    end

    # The cloud key generated ON THE GPU (tfhe_keygen_cloud_key) instead of by CloudKey(rng, secret_key) on the host
    # (api.jl:111-127): the TLWE key bits and a 64-bit seed come from `rng`, the bootstrap and keyswitch keys never
    # exist on the host.  The key material follows the library's Philox streams, not MersenneTwister's.
    function GpuCloudKey(rng::AbstractRNG, secret_key::SecretKey; device::Integer=0, devices=nothing)
        p = secret_key.params
        ctx = create_context(p, devices === nothing ? [device] : devices)
        lwe_bits = Int32.(secret_key.key.key)                                        # lwe.jl:11-17
        tlwe_bits = Int32.(rand(rng, Bool, p.tlwe_polynomial_degree, p.tlwe_mask_size))   # [N, k] = C-order [k][N]; tlwe.jl:15-20
        seed = rand(rng, UInt64)
        GC.@preserve lwe_bits tlwe_bits check(ctx, ccall((:tfhe_keygen_cloud_key, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Float64, Float64, UInt64, Ptr{Int32}, Ptr{Int32}),
            ctx, lwe_bits, tlwe_bits, p.bs_noise_stddev, p.ks_noise_stddev, seed, C_NULL, C_NULL))
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

# ---- multi-key (src/mk_api.jl:83-101, src/mk_gates.jl:7-12) ---------------------------------------------------
# MKBootstrapKey.key[j, i] (bit j of party i) :: MKTransformedTGswExpSample with spectra x[l, P], y[l, P], c0[l],
# c1[l] (src/mk_internals.jl:274-288, 442-461) -> complex128, C order [P][n][2lP + 2l][N/2], per (i, j) the polys
# x[p, q] at (p-1)*P + q, then y, then c0, then c1 (include/tfhe_mi355x.h).
function flatten_mk_spectra(bk, p::SchemeParameters, parties::Int)
    n, l, M = p.lwe_size, p.bs_decomp_length, p.tlwe_polynomial_degree ÷ 2
    per = 2 * l * parties + 2 * l
    spectra = Array{Complex{Float64}}(undef, M, per, n, parties)      # column-major: M fastest
    for i in 1:parties, j in 1:n
        s = bk.key[j, i]
        for pp in 1:l, q in 1:parties
            spectra[:, (pp - 1) * parties + q, j, i] .= s.x[pp, q].coeffs
            spectra[:, l * parties + (pp - 1) * parties + q, j, i] .= s.y[pp, q].coeffs
        end
        for pp in 1:l
            spectra[:, 2 * l * parties + pp, j, i] .= s.c0[pp].coeffs
            spectra[:, 2 * l * parties + l + pp, j, i] .= s.c1[pp].coeffs
        end
    end
    spectra
end

mutable struct GpuMKCloudKey
    params::SchemeParameters
    parties::Int
    ctx::Ptr{Cvoid}

    function GpuMKCloudKey(ck::MKCloudKey; device::Integer=0, devices=nothing)
        p, P = ck.params, ck.parties
        ctx = create_context(p, devices === nothing ? [device] : devices)
        spectra = flatten_mk_spectra(ck.bootstrap_key, p, P)
        GC.@preserve spectra check(ctx, ccall((:tfhe_mk_load_bootstrap_key_c128, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Complex{Float64}}, Int32), ctx, spectra, Int32(P)))
        # P single-key keyswitch keys back to back (src/mk_api.jl:97-98)
        flat = cat([flatten_keyswitch_key(ks, p.lwe_size) for ks in ck.keyswitch_key]...; dims=5)
        GC.@preserve flat check(ctx, ccall((:tfhe_mk_load_keyswitch_key, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Int32}, Int32), ctx, flat, Int32(P)))
        mck = new(p, P, ctx)
        finalizer(g -> ccall((:tfhe_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), g.ctx), mck)
        mck
    end
end

# MKLweSample (src/mk_internals.jl:6-18: a is n x P, one column per party) <-> flat Int32[P*n+1] = a[:,1]; a[:,2]; ...; b
function flatten(xs::AbstractVector{MKLweSample})
    n, P = size(xs[1].a)
    m = Array{Int32}(undef, n * P + 1, length(xs))
    for (g, x) in enumerate(xs)
        m[1:n*P, g] .= vec(x.a)                                        # column-major vec = party columns in order
        m[n * P + 1, g] = x.b
    end
    m
end

function mk_gate_nand(mck::GpuMKCloudKey, xs::AbstractVector{MKLweSample}, ys::AbstractVector{MKLweSample})
    n, P, B = mck.params.lwe_size, mck.parties, length(xs)
    fx, fy = flatten(xs), flatten(ys)
    out = Array{Int32}(undef, n * P + 1, B)
    GC.@preserve fx fy out check(mck.ctx, ccall((:tfhe_mk_gate_nand_batch, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64), mck.ctx, fx, fy, out, B))
    params = LweParams(n)
    # current_variance: 0.0 as the reference's own TODO leaves it (src/mk_internals.jl:94)
    [MKLweSample(params, reshape(out[1:n*P, g], n, P), out[n * P + 1, g], 0.) for g in 1:B]
end

mk_gate_nand(mck::GpuMKCloudKey, x::MKLweSample, y::MKLweSample) = mk_gate_nand(mck, [x], [y])[1]

end # module
