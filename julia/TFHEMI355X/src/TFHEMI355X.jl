# TFHEMI355X.jl — puts libtfhe_mi355x.so behind TFHE.jl's OWN gate functions.
#
# NOT EXECUTED IN THE BUILD IMAGE (no Julia there): kept literal so it can be reviewed by reading against
# include/tfhe_mi355x.h; tests/test_julia_shim.py checks every ccall against the header, that no TFHE name is exported
# without being imported from TFHE, that every exported gate has its batched `broadcasted` method, and every size(...)
# destructuring against the reference's comprehension order.  One command runs it where Julia exists (INTEGRATION.md §2):
#     julia --project=julia/TFHEMI355X -e 'using Pkg; Pkg.develop(path="<TFHE.jl checkout>"); Pkg.test()'
#
# The module defines NO function named gate_*: it `import`s TFHE's functions (src/TFHE.jl:34-46,61) and adds methods
# whose cloud-key argument is a GpuCloudKey / GpuMKCloudKey.  After
#
#     using TFHE, Random
#     using TFHEMI355X                                        # the package julia/TFHEMI355X (Project.toml: deps TFHE, Random); exports only GpuCloudKey, GpuMKCloudKey, GpuLweArray, ...
#     rng = MersenneTwister(123)
#     secret_key, cloud_key = make_key_pair(rng)
#     gck = GpuCloudKey(cloud_key)                            # flattens + uploads the keys once (device 0)
#
# the reference's caller lines run unchanged with `gck` where they say `cloud_key`:
#
#     cresult = gate_xor.(gck, ciphertext1, ciphertext2)      # docs/src/manual.md:35 — ONE tfhe_gates_batch call
#     tmp = gate_xnor(gck, a, b); gate_mux(gck, tmp, lsb_carry, a)          # examples/tutorial.jl:44-45
#     tmps1 = gate_constant(gck, false)                       # examples/tutorial.jl:54
#     [gate_mux(gck, tmps1, b[i], a[i]) for i in 1:nb_bits]   # examples/tutorial.jl:62 (or gate_mux.(gck, tmps1, b, a): one call)
#     enc_out = mk_gate_nand(mck, enc_mess1, enc_mess2)       # test/runtests.jl:95, mck = GpuMKCloudKey(cloud_key)
#
# (examples/tutorial.jl annotates its own helper functions `ck::CloudKey`; to pass a GpuCloudKey through them the
# annotation has to be dropped or widened — CloudKey is a concrete struct, nothing can subtype it.)
#
#     gck8 = GpuCloudKey(cloud_key; devices=0:7)              # keys replicated on 8 GPUs, every batch call split over them
#     d1 = upload(gck, ciphertext1); d2 = upload(gck, ciphertext2)          # GpuLweArray: ciphertexts resident on the GPU
#     d3 = gate_and.(gck, gate_xor.(gck, d1, d2), d1)         # stays on the device between gates (tfhe_gates_level)
#     result = download(d3)                                   # Vector{LweSample}
module TFHEMI355X

using TFHE
import TFHE: gate_nand, gate_or, gate_and, gate_xor, gate_xnor, gate_not, gate_constant,
             gate_nor, gate_andny, gate_andyn, gate_orny, gate_oryn, gate_mux, mk_gate_nand
using TFHE: LweSample, LweParams, CloudKey, SecretKey, SchemeParameters, MKCloudKey, MKLweSample
using Random: AbstractRNG, RandomDevice
import Base.Broadcast: broadcastable, broadcasted

export GpuCloudKey, GpuMKCloudKey, GpuLweArray, gates_batch, gates_batch_async, PendingGates, upload, download

# the shared library as this repository builds it (make -C tfhe.jl_amd/csrc), or wherever TFHE_MI355X_LIB points
const LIB = get(ENV, "TFHE_MI355X_LIB", joinpath(@__DIR__, "..", "..", "..", "tfhe.jl_amd", "lib", "libtfhe_mi355x.so"))

# ABI check at load time (include/tfhe_mi355x.h: TFHE_MI355X_ABI_VERSION).  A NEGATIVE answer is a development build of the
# library (-DTFHE_EXPERIMENT, csrc/experiment.hpp: in-kernel stamps, environment-variable overrides): refused unless the user
# asked for it.
const ABI_VERSION = Int32(7)
function __init__()
    v = ccall((:tfhe_abi_version, LIB), Int32, ())
    if v < 0 && get(ENV, "TFHE_MI355X_ALLOW_EXPERIMENT", "") == ""
        error("$LIB is a development build (ABI version $v): set TFHE_MI355X_ALLOW_EXPERIMENT=1 to load it knowingly")
    end
    abs(v) == ABI_VERSION || error("$LIB has ABI version $v, this package needs $ABI_VERSION: rebuild it (make -C tfhe.jl_amd/csrc)")
end

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

# One caller at a time per context (include/tfhe_mi355x.h, "Threading"): the library answers an overlapping call from another
# thread with TFHE_ERR_STATE; Julia tasks sharing a key therefore take the context's lock around every call (re-entrant: the
# scalar gate methods call the batch methods).  The lock also covers tfhe_last_error, whose message belongs to the owner.
const CTX_LOCKS = Dict{Ptr{Cvoid}, ReentrantLock}()
const CTX_LOCKS_GUARD = ReentrantLock()
ctx_lock(ctx::Ptr{Cvoid}) = lock(CTX_LOCKS_GUARD) do
    get!(() -> ReentrantLock(), CTX_LOCKS, ctx)
end
macro locked(ctx, ex)
    quote
        local lk = ctx_lock($(esc(ctx)))
        lock(lk)
        try
            $(esc(ex))
        finally
            unlock(lk)
        end
    end
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
    warn_if_inexact(ctxref[], p)
    ctxref[]
end

# tfhe_get_option "exact_domain" (ABI v7): 0 = the parameter set is outside what a Float64 transform computes exactly — the
# reference warns about the same limit (src/polynomials.jl:135-144); said once per parameter set
const WARNED_INEXACT = Set{Any}()
function warn_if_inexact(ctx::Ptr{Cvoid}, p::SchemeParameters)
    opt(name) = (v = Ref{Int64}(0); ccall((:tfhe_get_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Ref{Int64}), ctx, name, v) == 0 ? v[] : Int64(-1))
    opt("exact_domain") == 0 || return
    key = (p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length, p.bs_log2_base, p.max_parties)
    key in WARNED_INEXACT && return
    push!(WARNED_INEXACT, key)
    @warn "TFHE parameter set is outside the Float64 exactness domain: result words may differ from the exact negacyclic product" predicted_rounding_margin = opt("exact_margin_x1e6") / 1e6 log2_worst_case_magnitude = opt("exact_bound_log2_x1000") / 1e3
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

"""
    GpuCloudKey(ck::CloudKey; device=0, devices=nothing, wires=65536)
    GpuCloudKey(rng, secret_key::SecretKey; device=0, devices=nothing, wires=65536)

A `CloudKey` resident on one MI355X (`device`) or replicated on several (`devices=0:7`).  Pass it to TFHE's own
`gate_*` functions in place of the `CloudKey`.  `wires` is the capacity (in ciphertexts) of the device-resident table
behind [`GpuLweArray`](@ref); it is allocated on first use.
"""
mutable struct GpuCloudKey
    params::SchemeParameters
    ctx::Ptr{Cvoid}
    # device-resident ciphertexts (tfhe_wires_*): rows of one table, handed out here, handed back by GpuLweArray finalizers
    wire_capacity::Int
    wire_ready::Bool                       # tfhe_wires_alloc done
    wire_next::Int
    wire_free::Vector{Int32}
    wire_returned::Vector{Vector{Int32}}

    # the finalizer is attached before anything that can throw: a failed key load does not leak the context
    function GpuCloudKey(p::SchemeParameters, devices, wires::Integer)
        ctx = create_context(p, devices)
        gck = new(p, ctx, Int(wires), false, 0, Int32[], Vector{Vector{Int32}}())
        finalizer(destroy!, gck)
        # no per-phase timing events: nothing in this module reads them, and every record keeps the stream's next kernel
        # waiting ~5 us (six per circuit level; include/tfhe_mi355x.h, tfhe_set_option)
        @locked ctx check(ctx, ccall((:tfhe_set_option, LIB), Int32, (Ptr{Cvoid}, Cstring, Int64), ctx, "timing_events", Int64(0)))
        gck
    end
end

function destroy!(g)
    if g.ctx != C_NULL
        ccall((:tfhe_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), g.ctx)    # (its entry in CTX_LOCKS stays: this may run as a finalizer, which must not take locks; a reused address finds a valid lock)
        g.ctx = C_NULL
    end
    nothing
end

function GpuCloudKey(ck::CloudKey; device::Integer=0, devices=nothing, wires::Integer=65536)
    p = ck.params
    gck = GpuCloudKey(p, devices === nothing ? [device] : devices, wires)
    try
        spectra = flatten_bootstrap_spectra(ck.bootstrap_key, p)
        GC.@preserve spectra @locked gck.ctx check(gck.ctx, ccall((:tfhe_load_bootstrap_key_c128, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Complex{Float64}}), gck.ctx, spectra))
        flat = flatten_keyswitch_key(ck.keyswitch_key, p.lwe_size)
        GC.@preserve flat @locked gck.ctx check(gck.ctx, ccall((:tfhe_load_keyswitch_key, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Int32}), gck.ctx, flat))
    catch
        destroy!(gck)
        rethrow()
    end
    gck
end

# The cloud key generated ON THE GPU (tfhe_keygen_cloud_key) instead of by CloudKey(rng, secret_key) on the host
# (api.jl:111-127): the TLWE key bits and a seed of six 32-bit words come from `rng` (for real keys: a cryptographic
# generator such as RandomDevice()), the bootstrap and keyswitch keys never exist on the host.  The key material follows
# the library's Philox streams, not MersenneTwister's.  Four of the seed words key the noise and are as secret as the
# secret key (they regenerate the noise of every key row): the seed is not stored anywhere.
function GpuCloudKey(rng::AbstractRNG, secret_key::SecretKey; device::Integer=0, devices=nothing, wires::Integer=65536,
                     noise_seed=nothing)     # (tests only: four fixed words instead of RandomDevice())
    p = secret_key.params
    gck = GpuCloudKey(p, devices === nothing ? [device] : devices, wires)
    try
        lwe_bits = Int32.(secret_key.key.key)                                        # lwe.jl:11-17
        tlwe_bits = Int32.(rand(rng, Bool, p.tlwe_polynomial_degree, p.tlwe_mask_size))   # [N, k] = C-order [k][N]; tlwe.jl:15-20
        # words 1-2 key the public masks (from `rng`: reproducible), words 3-6 (128 bits) key the noise: secret, drawn from the
        # operating system's generator whatever `rng` is (Philox then only expands that secret), discarded here
        seed = vcat(rand(rng, UInt32, 2), noise_seed === nothing ? rand(RandomDevice(), UInt32, 4) : UInt32.(noise_seed))
        length(seed) == 6 || error("GpuCloudKey: noise_seed must be four 32-bit words")
        GC.@preserve lwe_bits tlwe_bits seed @locked gck.ctx check(gck.ctx, ccall((:tfhe_keygen_cloud_key, LIB), Int32,
            (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Float64, Float64, Ptr{UInt32}, Ptr{Int32}, Ptr{Int32}),
            gck.ctx, lwe_bits, tlwe_bits, p.bs_noise_stddev, p.ks_noise_stddev, seed, C_NULL, C_NULL))
    catch
        destroy!(gck)
        rethrow()
    end
    gck
end

# a GpuCloudKey is a scalar under broadcasting, as CloudKey is (src/api.jl:130)
broadcastable(g::GpuCloudKey) = Ref(g)

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

# ---- device-resident ciphertexts ------------------------------------------------------------------------------------
"""
    GpuLweArray

A vector of LWE samples resident on the GPU (rows of the key's wire table, tfhe_wires_*).  `upload(gck, xs)` makes
one, `download(d)` brings it back as `Vector{LweSample}`; gates on GpuLweArrays run through `tfhe_gates_level` and
return GpuLweArrays, so a circuit's intermediate ciphertexts never cross PCIe and are never re-flattened.
"""
mutable struct GpuLweArray
    key::GpuCloudKey
    rows::Vector{Int32}            # wire indices (0-based)

    function GpuLweArray(key::GpuCloudKey, rows::Vector{Int32})
        d = new(key, rows)
        # a finalizer must not call into the allocator: it only queues the rows; alloc_rows! takes them back
        finalizer(a -> push!(a.key.wire_returned, a.rows), d)
        d
    end
end

Base.length(d::GpuLweArray) = length(d.rows)

function alloc_rows!(g::GpuCloudKey, count::Int)
    if !g.wire_ready                                                       # first use: allocate the table
        @locked g.ctx check(g.ctx, ccall((:tfhe_wires_alloc, LIB), Int32, (Ptr{Cvoid}, Int64), g.ctx, g.wire_capacity))
        g.wire_ready = true
    end
    returned = g.wire_returned
    g.wire_returned = Vector{Vector{Int32}}()
    for r in returned
        append!(g.wire_free, r)
    end
    rows = Vector{Int32}(undef, count)
    reused = min(count, length(g.wire_free))
    for i in 1:reused
        rows[i] = pop!(g.wire_free)
    end
    fresh = count - reused
    if g.wire_next + fresh > g.wire_capacity
        append!(g.wire_free, rows[1:reused])
        error("GpuLweArray: wire table full ($(g.wire_capacity) ciphertexts); construct the GpuCloudKey with a larger `wires`")
    end
    for i in 1:fresh
        rows[reused + i] = g.wire_next + i - 1
    end
    g.wire_next += fresh
    rows
end

"""
    upload(gck, xs::AbstractVector{LweSample}) -> GpuLweArray
"""
function upload(g::GpuCloudKey, xs::AbstractVector{LweSample})
    rows = alloc_rows!(g, length(xs))
    isempty(xs) && return GpuLweArray(g, rows)
    flat = flatten(xs)
    first_fresh = isempty(rows) ? 0 : rows[1]
    contiguous = rows == collect(Int32, first_fresh:(first_fresh + length(rows) - 1))
    if contiguous
        GC.@preserve flat @locked g.ctx check(g.ctx, ccall((:tfhe_wires_upload, LIB), Int32,
            (Ptr{Cvoid}, Int64, Int64, Ptr{Int32}), g.ctx, first_fresh, length(rows), flat))
    else                                                                 # recycled rows: one copy per row
        n1 = size(flat, 1)
        for (i, r) in enumerate(rows)
            GC.@preserve flat @locked g.ctx check(g.ctx, ccall((:tfhe_wires_upload, LIB), Int32,
                (Ptr{Cvoid}, Int64, Int64, Ptr{Int32}), g.ctx, r, 1, pointer(flat, (i - 1) * n1 + 1)))
        end
    end
    GpuLweArray(g, rows)
end

"""
    download(d::GpuLweArray) -> Vector{LweSample}
"""
function download(d::GpuLweArray)
    g = d.key
    out = Array{Int32}(undef, g.params.lwe_size + 1, length(d))
    GC.@preserve out @locked g.ctx check(g.ctx, ccall((:tfhe_wires_gather, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Int32}, Int64, Ptr{Int32}), g.ctx, d.rows, length(d), out))
    unflatten(out, LweParams(g.params.lwe_size))
end

# ---- the batch calls ------------------------------------------------------------------------------------------------
const LweVec = AbstractVector{LweSample}
const LweOperand = Union{LweSample, LweVec, GpuLweArray}

operand_length(x::LweSample) = 1
operand_length(x) = length(x)

# the common length of the vector operands (scalars repeat, as under broadcasting)
function batch_length(operands)
    B = 1
    for x in operands
        x isa LweSample && continue
        L = operand_length(x)
        (B == 1 || L == B || L == 1) || throw(DimensionMismatch("gate operands of lengths $B and $L"))
        B = max(B, L)
    end
    B
end

expand(x::LweSample, B) = fill(x, B)
expand(xs::LweVec, B) = length(xs) == B ? xs : fill(xs[1], B)

"""
    gates_batch(gck, opcodes, xs, ys=nothing, zs=nothing)

`length(opcodes)` independent gates, one opcode per gate, in one GPU call (tfhe_gates_batch).
"""
function gates_batch(gck::GpuCloudKey, opcodes::Vector{UInt8}, xs, ys=nothing, zs=nothing)
    B = length(opcodes)
    params = LweParams(gck.params.lwe_size)
    B == 0 && return LweSample[]
    fx = xs === nothing ? nothing : flatten(xs)
    fy = ys === nothing ? nothing : flatten(ys)
    fz = zs === nothing ? nothing : flatten(zs)
    out = Array{Int32}(undef, gck.params.lwe_size + 1, B)
    ptr(a) = a === nothing ? Ptr{Int32}(C_NULL) : pointer(a)
    GC.@preserve fx fy fz out opcodes @locked gck.ctx check(gck.ctx, ccall((:tfhe_gates_batch, LIB), Int32,
        (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64),
        gck.ctx, opcodes, ptr(fx), ptr(fy), ptr(fz), out, B))
    unflatten(out, params)
end

# page-locked Int32 matrix (tfhe_host_alloc): the copies of a streamed batch are then single DMA transfers that overlap kernels
function pinned_matrix(rows::Int, cols::Int)
    p = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:tfhe_host_alloc, LIB), Int32, (Csize_t, Ptr{Ptr{Cvoid}}), Csize_t(4 * rows * cols), p)
    rc == 0 || error("tfhe_mi355x: tfhe_host_alloc failed (", rc, ")")
    unsafe_wrap(Array, Ptr{Int32}(p[]), (rows, cols); own=false)
end
release_pinned(a::Array{Int32}) = ccall((:tfhe_host_free, LIB), Cvoid, (Ptr{Cvoid},), pointer(a))

"""
    t = gates_batch_async(gck, opcodes, xs, ys=nothing, zs=nothing)  ->  PendingGates
    fetch(t)                                                         ->  Vector{LweSample}

Streaming form of `gates_batch` (tfhe_gates_batch_submit / tfhe_gates_batch_wait): the call returns once the batch is
enqueued; up to two batches run at a time, the upload of one under the kernels of the other, so a caller that feeds
batch after batch pays no PCIe time in the steady state.  Operands are staged in page-locked buffers that `fetch` frees.
"""
mutable struct PendingGates
    key::GpuCloudKey
    ticket::Int32
    buffers::Vector{Array{Int32}}     # page-locked operand copies + the result, alive until fetch
    out::Array{Int32}
    done::Bool
end

function gates_batch_async(gck::GpuCloudKey, opcodes::Vector{UInt8}, xs, ys=nothing, zs=nothing)
    B = length(opcodes)
    B > 0 || error("gates_batch_async: empty batch")
    n1 = gck.params.lwe_size + 1
    stage(v) = v === nothing ? nothing : copyto!(pinned_matrix(n1, B), flatten(v))
    fx, fy, fz = stage(xs), stage(ys), stage(zs)
    out = pinned_matrix(n1, B)
    ptr(a) = a === nothing ? Ptr{Int32}(C_NULL) : pointer(a)
    ticket = Ref{Int32}(-1)
    bufs = Array{Int32}[b for b in (fx, fy, fz, out) if b !== nothing]
    rc = @locked gck.ctx ccall((:tfhe_gates_batch_submit, LIB), Int32,
        (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64, Ptr{Int32}),
        gck.ctx, opcodes, ptr(fx), ptr(fy), ptr(fz), out, B, ticket)
    if rc != 0
        foreach(release_pinned, bufs)
        @locked gck.ctx check(gck.ctx, rc)
    end
    t = PendingGates(gck, ticket[], bufs, out, false)
    # a PendingGates that is dropped (or whose fetch is never reached) must not leak its page-locked buffers: wait for the
    # batch, then free them (a finalizer may call into C; it must not allocate Julia objects, and this one does not)
    finalizer(abandon!, t)
    t
end

# Runs as a finalizer: it cannot take the context's lock, and another task may be inside a gate call on the same context at
# this very moment.  tfhe_ctx_synchronize (ABI v7) is the one entry point made for that: it takes no ownership of the context,
# touches none of its state and returns once everything queued so far — this batch's copies included — has completed.  The
# page-locked buffers go back ONLY after it returned TFHE_OK: on any other answer they are leaked rather than freed under a DMA
# that may still be running.  (Until ABI v6 this called tfhe_gates_batch_wait, which the library's one-caller-at-a-time guard
# answered with TFHE_ERR_STATE at once while another task was inside a call — and the buffers were freed regardless.)
function abandon!(t::PendingGates)
    t.done && return nothing
    drained = t.key.ctx == C_NULL ||      # (the context is gone: tfhe_ctx_destroy synchronised its streams)
              ccall((:tfhe_ctx_synchronize, LIB), Int32, (Ptr{Cvoid},), t.key.ctx) == 0
    drained && foreach(release_pinned, t.buffers)
    empty!(t.buffers)
    t.done = true
    nothing
end

function Base.fetch(t::PendingGates)
    t.done && error("PendingGates: already fetched")
    rc = @locked t.key.ctx ccall((:tfhe_gates_batch_wait, LIB), Int32, (Ptr{Cvoid}, Int32), t.key.ctx, t.ticket)
    # a failed wait says nothing about the copies still in flight: drain the context (tfhe_ctx_synchronize needs no lock and
    # cannot be refused) before the buffers go back; if even that fails they are leaked, not freed under a live DMA
    drained = rc == 0 || ccall((:tfhe_ctx_synchronize, LIB), Int32, (Ptr{Cvoid},), t.key.ctx) == 0
    res = rc == 0 ? unflatten(copy(t.out), LweParams(t.key.params.lwe_size)) : nothing
    drained && foreach(release_pinned, t.buffers)
    empty!(t.buffers)
    t.done = true
    check(t.key.ctx, rc)
    res
end

# one level of B gates on device-resident operands (tfhe_gates_level); host samples among them are uploaded first
function gates_on_device(g::GpuCloudKey, op::UInt8, operands)
    B = batch_length(operands)
    devs = map(operands) do x
        d = x isa GpuLweArray ? x : upload(g, x isa LweSample ? [x] : x)
        d.key === g || error("GpuLweArray belongs to another GpuCloudKey")
        d
    end
    index(d) = length(d) == B ? d.rows : fill(d.rows[1], B)
    ia = index(devs[1])
    ib = length(devs) >= 2 ? index(devs[2]) : Int32[]
    ic = length(devs) >= 3 ? index(devs[3]) : Int32[]
    out = alloc_rows!(g, B)
    opcodes = fill(op, B)
    ptr(a) = isempty(a) ? Ptr{Int32}(C_NULL) : pointer(a)
    GC.@preserve opcodes ia ib ic out devs @locked g.ctx check(g.ctx, ccall((:tfhe_gates_level, LIB), Int32,
        (Ptr{Cvoid}, Ptr{UInt8}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64),
        g.ctx, opcodes, ptr(ia), ptr(ib), ptr(ic), out, B))
    GpuLweArray(g, out)
end

# One gate kind over scalar / vector / device operands: all LweSample -> a LweSample (batch of one); any vector ->
# Vector{LweSample} from ONE tfhe_gates_batch; any GpuLweArray -> a GpuLweArray (nothing leaves the device).
function run_gate(g::GpuCloudKey, op::UInt8, operands...)
    any(x -> x isa GpuLweArray, operands) && return gates_on_device(g, op, operands)
    B = batch_length(operands)
    vecs = map(x -> expand(x, B), operands)
    res = gates_batch(g, fill(op, B), vecs...)
    all(x -> x isa LweSample, operands) ? res[1] : res
end

# Methods ON TFHE's functions (src/gates.jl:15-153), and the batched forms of their broadcast: gate_xor.(gck, c1, c2)
# (docs/src/manual.md:35) lowers to broadcasted(gate_xor, gck, c1, c2), which these methods answer with one GPU call
# instead of length(c1) single-gate launches.
for (fn, op) in ((:gate_nand, NAND), (:gate_or, OR), (:gate_and, AND), (:gate_xor, XOR),
                 (:gate_xnor, XNOR), (:gate_nor, NOR), (:gate_andny, ANDNY), (:gate_andyn, ANDYN),
                 (:gate_orny, ORNY), (:gate_oryn, ORYN))
    @eval begin
        $fn(g::GpuCloudKey, x::LweOperand, y::LweOperand) = run_gate(g, $op, x, y)
        broadcasted(::typeof($fn), g::GpuCloudKey, x::LweOperand, y::LweOperand) = run_gate(g, $op, x, y)
        broadcasted(::typeof($fn), g::Base.RefValue{GpuCloudKey}, x::LweOperand, y::LweOperand) = run_gate(g[], $op, x, y)
    end
end

# src/gates.jl:163-177
gate_mux(g::GpuCloudKey, x::LweOperand, y::LweOperand, z::LweOperand) = run_gate(g, MUX, x, y, z)
broadcasted(::typeof(gate_mux), g::GpuCloudKey, x::LweOperand, y::LweOperand, z::LweOperand) = run_gate(g, MUX, x, y, z)
broadcasted(::typeof(gate_mux), g::Base.RefValue{GpuCloudKey}, x::LweOperand, y::LweOperand, z::LweOperand) =
    run_gate(g[], MUX, x, y, z)

# not bootstrapped (src/gates.jl:76-93): on host samples no device round trip is needed
gate_not(g::GpuCloudKey, x::LweSample) = -x
gate_not(g::GpuCloudKey, xs::LweVec) = [-x for x in xs]
gate_not(g::GpuCloudKey, d::GpuLweArray) = gates_on_device(g, NOT, (d,))
broadcasted(::typeof(gate_not), g::GpuCloudKey, x::LweOperand) = gate_not(g, x)
broadcasted(::typeof(gate_not), g::Base.RefValue{GpuCloudKey}, x::LweOperand) = gate_not(g[], x)

gate_constant(g::GpuCloudKey, value::Bool) =
    TFHE.lwe_noiseless_trivial(TFHE.encode_message(value ? 1 : -1, 8), LweParams(g.params.lwe_size))
broadcasted(::typeof(gate_constant), g::GpuCloudKey, values::AbstractVector{Bool}) = [gate_constant(g, v) for v in values]
broadcasted(::typeof(gate_constant), g::Base.RefValue{GpuCloudKey}, values::AbstractVector{Bool}) =
    [gate_constant(g[], v) for v in values]

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

"""
    GpuMKCloudKey(ck::MKCloudKey; device=0, devices=nothing)

An `MKCloudKey` resident on the GPU(s); pass it to TFHE's own `mk_gate_nand` in place of the `MKCloudKey`.
"""
mutable struct GpuMKCloudKey
    params::SchemeParameters
    parties::Int
    ctx::Ptr{Cvoid}

    function GpuMKCloudKey(ck::MKCloudKey; device::Integer=0, devices=nothing)
        p, P = ck.params, ck.parties
        ctx = create_context(p, devices === nothing ? [device] : devices)
        mck = new(p, P, ctx)
        finalizer(destroy!, mck)
        try
            spectra = flatten_mk_spectra(ck.bootstrap_key, p, P)
            GC.@preserve spectra @locked ctx check(ctx, ccall((:tfhe_mk_load_bootstrap_key_c128, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Complex{Float64}}, Int32), ctx, spectra, Int32(P)))
            # P single-key keyswitch keys back to back (src/mk_api.jl:97-98)
            flat = cat([flatten_keyswitch_key(ks, p.lwe_size) for ks in ck.keyswitch_key]...; dims=5)
            GC.@preserve flat @locked ctx check(ctx, ccall((:tfhe_mk_load_keyswitch_key, LIB), Int32,
                (Ptr{Cvoid}, Ptr{Int32}, Int32), ctx, flat, Int32(P)))
        catch
            destroy!(mck)
            rethrow()
        end
        mck
    end
end

broadcastable(m::GpuMKCloudKey) = Ref(m)

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

const MKVec = AbstractVector{MKLweSample}

function mk_nand_batch(mck::GpuMKCloudKey, xs::MKVec, ys::MKVec)
    length(xs) == length(ys) || throw(DimensionMismatch("mk_gate_nand operands of lengths $(length(xs)) and $(length(ys))"))
    n, P, B = mck.params.lwe_size, mck.parties, length(xs)
    fx, fy = flatten(xs), flatten(ys)
    out = Array{Int32}(undef, n * P + 1, B)
    GC.@preserve fx fy out @locked mck.ctx check(mck.ctx, ccall((:tfhe_mk_gate_nand_batch, LIB), Int32,
        (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{Int32}, Int64), mck.ctx, fx, fy, out, B))
    params = LweParams(n)
    # current_variance: 0.0 as the reference's own TODO leaves it (src/mk_internals.jl:94)
    [MKLweSample(params, reshape(out[1:n*P, g], n, P), out[n * P + 1, g], 0.) for g in 1:B]
end

# methods on TFHE's mk_gate_nand (src/mk_gates.jl:7-12): scalar, vectors, and the batched form of its broadcast
mk_gate_nand(mck::GpuMKCloudKey, x::MKLweSample, y::MKLweSample) = mk_nand_batch(mck, [x], [y])[1]
mk_gate_nand(mck::GpuMKCloudKey, xs::MKVec, ys::MKVec) = mk_nand_batch(mck, xs, ys)
broadcasted(::typeof(mk_gate_nand), mck::GpuMKCloudKey, xs::MKVec, ys::MKVec) = mk_nand_batch(mck, xs, ys)
broadcasted(::typeof(mk_gate_nand), mck::Base.RefValue{GpuMKCloudKey}, xs::MKVec, ys::MKVec) = mk_nand_batch(mck[], xs, ys)

end # module
