# mint_fixtures.jl — mints golden vectors FROM THE REAL REFERENCE (nucypher/TFHE.jl) in this repo's container format.
#
# NOT EXECUTED IN THE BUILD IMAGE (no Julia there).  Anyone with Julia + TFHE.jl can close the "parity unpinned" gap:
#
#     julia --project=julia/TFHEMI355X julia/TFHEMI355X/scripts/mint_fixtures.jl [outdir = tests/golden] [lwe_size = 16]
# (after `Pkg.develop(path="<TFHE.jl checkout>")` in that project, INTEGRATION.md §2; test/runtests.jl calls it too)
#
# writes  ref_gates80.tfhe   13 gate kinds x all input combinations, tfhe_parameters_80, seed 123 (test/runtests.jl:26-40)
#         ref_gates128.tfhe  NAND / MUX truth tables, tfhe_parameters_128                     (test/runtests.jl:43-57)
#         ref_mk2.tfhe       2-party multi-key NAND x 10                                       (test/runtests.jl:60-100)
#         ref_gates_n512.tfhe, ref_gates_n4096.tfhe   NAND / XOR / MUX truth tables on sets the reference ships no constructor for
#                            but accepts (SchemeParameters is a positional struct, src/api.jl:4-21): N = 512 with
#                            tfhe_parameters_80's other fields, N = 4096 with tfhe_parameters_128's — the engine's any-N kernels
#         ref_mk2_n512.tfhe  2-party multi-key NAND x 10 at N = 512
# By default the LWE dimension is cut to 16 (every other parameter as shipped) so that a file is 1-5 MB and can be
# committed; pass lwe_size = 0 for the full-size sets (82 / 100 / 300 MB).  The blind rotation then has 16 steps instead
# of 500 / 630: the same code path, a shorter loop.
# `python -m pytest tests/test_golden.py` then checks the oracle (CPU) and the HIP engine (-m gpu) against the
# reference's own output words, bit for bit (it picks up every tests/golden/ref_*.tfhe present).  Only the exported API computes anything (make_key_pair, encrypt, gate_*,
# SharedKey, CloudKeyPart, MKCloudKey, mk_encrypt, mk_gate_nand: src/TFHE.jl:24-61); keys are flattened exactly as the
# GPU shim does it (julia/TFHEMI355X/src/TFHEMI355X.jl), so a fixture also pins the shim's layout.
#
# Container (tfhe.jl_amd/serialize.py): magic "TFHEMI355X\0" | version u32 | n_sections u32 | per section: name[16] |
# dtype u32 (0 Int32, 1 Float64, 2 Complex{Float64}, 3 UInt8) | ndim u32 | shape u64[ndim] (C order) | raw data.
# A Julia array of size (d1, ..., dk) is, byte for byte, a C-order array of shape (dk, ..., d1).
using Random
using TFHE

using TFHEMI355X: flatten, flatten_bootstrap_spectra, flatten_keyswitch_key, flatten_mk_spectra,
                   NAND, OR, AND, XOR, XNOR, NOT, NOR, ANDNY, ANDYN, ORNY, ORYN, MUX, CONST0, CONST1

dtype_code(::Type{Int32}) = UInt32(0)
dtype_code(::Type{Float64}) = UInt32(1)
dtype_code(::Type{Complex{Float64}}) = UInt32(2)
dtype_code(::Type{UInt8}) = UInt32(3)

function write_sections(path, sections)
    open(path, "w") do f
        write(f, Vector{UInt8}("TFHEMI355X"), UInt8(0))
        write(f, UInt32(1), UInt32(length(sections)))
        for (name, arr) in sections
            nb = Vector{UInt8}(name)
            @assert length(nb) <= 16
            write(f, nb, zeros(UInt8, 16 - length(nb)))
            write(f, dtype_code(eltype(arr)), UInt32(ndims(arr)))
            for d in reverse(size(arr))
                write(f, UInt64(d))
            end
            write(f, arr)
        end
    end
    println("wrote ", path)
end

params_vec(p) = Int32[p.lwe_size, p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length, p.bs_log2_base,
                      p.ks_decomp_length, p.ks_log2_base, p.max_parties]
noise_vec(p) = Float64[p.lwe_noise_stddev, p.bs_noise_stddev, p.ks_noise_stddev]

# every gate kind over all input combinations, evaluated by the reference
function gate_cases(rng, secret_key, cloud_key, kinds)
    two = Dict(NAND => gate_nand, OR => gate_or, AND => gate_and, XOR => gate_xor, XNOR => gate_xnor, NOR => gate_nor,
               ANDNY => gate_andny, ANDYN => gate_andyn, ORNY => gate_orny, ORYN => gate_oryn)
    ops = UInt8[]; xs = LweSample[]; ys = LweSample[]; zs = LweSample[]; outs = LweSample[]; plain = UInt8[]
    for op in kinds, bx in (false, true), by in (false, true), bz in (false, true)
        x, y, z = encrypt(rng, secret_key, bx), encrypt(rng, secret_key, by), encrypt(rng, secret_key, bz)
        out = if haskey(two, op)
            two[op](cloud_key, x, y)
        elseif op == MUX
            gate_mux(cloud_key, x, y, z)
        elseif op == NOT
            gate_not(cloud_key, x)
        else
            gate_constant(cloud_key, op == CONST1)
        end
        push!(ops, op); push!(xs, x); push!(ys, y); push!(zs, z); push!(outs, out)
        push!(plain, UInt8(decrypt(secret_key, out)))
    end
    ops, xs, ys, zs, outs, plain
end

function mint_single(path, params, kinds; seed=123)
    rng = MersenneTwister(seed)                                     # test/runtests.jl:27
    secret_key, cloud_key = make_key_pair(rng, params)
    ops, xs, ys, zs, outs, plain = gate_cases(rng, secret_key, cloud_key, kinds)
    write_sections(path, [
        "params" => params_vec(params), "noise" => noise_vec(params),
        "bk_spectra" => flatten_bootstrap_spectra(cloud_key.bootstrap_key, params),
        "keyswitch_key" => flatten_keyswitch_key(cloud_key.keyswitch_key, params.lwe_size),
        "lwe_key" => Int32.(secret_key.key.key),
        "ops" => ops, "in0" => flatten(xs), "in1" => flatten(ys), "in2" => flatten(zs), "out" => flatten(outs),
        "plain" => plain])
end

function mint_mk(path, params, parties; trials=10)
    rng = MersenneTwister(321)
    secret_keys = [SecretKey(rng, params) for i in 1:parties]       # test/runtests.jl:69-79
    shared_key = SharedKey(rng, params)
    ck_parts = [CloudKeyPart(rng, sk, shared_key) for sk in secret_keys]
    cloud_key = MKCloudKey(ck_parts)
    xs = [mk_encrypt(rng, secret_keys, rand(rng, Bool)) for t in 1:trials]
    ys = [mk_encrypt(rng, secret_keys, rand(rng, Bool)) for t in 1:trials]
    outs = [mk_gate_nand(cloud_key, x, y) for (x, y) in zip(xs, ys)]
    plain = UInt8[mk_decrypt(secret_keys, o) for o in outs]
    n = params.lwe_size
    ks = cat([flatten_keyswitch_key(k, n) for k in cloud_key.keyswitch_key]...; dims=5)
    keys = Array{Int32}(undef, n, parties)
    for i in 1:parties
        keys[:, i] .= secret_keys[i].key.key
    end
    write_sections(path, [
        "params" => params_vec(params), "noise" => noise_vec(params), "parties" => Int32[parties],
        "mk_spectra" => flatten_mk_spectra(cloud_key.bootstrap_key, params, parties),
        "mk_keyswitch_key" => ks, "lwe_keys" => keys,
        "in0" => flatten(xs), "in1" => flatten(ys), "out" => flatten(outs), "plain" => plain])
end

# the shipped parameter set with another LWE dimension (SchemeParameters is a plain positional struct, src/api.jl:4-21)
with_lwe_size(p, n) = n == 0 ? p : TFHE.SchemeParameters(
    n, p.lwe_noise_stddev, p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length, p.bs_log2_base,
    p.bs_noise_stddev, p.ks_decomp_length, p.ks_log2_base, p.ks_noise_stddev, p.max_parties)

outdir = length(ARGS) >= 1 ? ARGS[1] : joinpath(@__DIR__, "..", "..", "..", "tests", "golden")
lwe_size = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 16
mkpath(outdir)
mint_single(joinpath(outdir, "ref_gates80.tfhe"), with_lwe_size(tfhe_parameters_80(), lwe_size),
            [NAND, OR, AND, XOR, XNOR, NOT, NOR, ANDNY, ANDYN, ORNY, ORYN, MUX, CONST0, CONST1])
mint_single(joinpath(outdir, "ref_gates128.tfhe"), with_lwe_size(tfhe_parameters_128(), lwe_size), [NAND, MUX])
mint_mk(joinpath(outdir, "ref_mk2.tfhe"), with_lwe_size(mktfhe_parameters_2party, lwe_size), 2)

# parameter sets outside the shipped polynomial degree (round 5: the engine accepts every power-of-two N; these pin its any-N
# kernels — csrc/kernels_anyn.hpp — against the reference too)
with_degree(p, N) = TFHE.SchemeParameters(
    p.lwe_size, p.lwe_noise_stddev, N, p.tlwe_mask_size, p.bs_decomp_length, p.bs_log2_base,
    p.bs_noise_stddev, p.ks_decomp_length, p.ks_log2_base, p.ks_noise_stddev, p.max_parties)
mint_single(joinpath(outdir, "ref_gates_n512.tfhe"), with_degree(with_lwe_size(tfhe_parameters_80(), lwe_size), 512), [NAND, XOR, MUX])
mint_single(joinpath(outdir, "ref_gates_n4096.tfhe"), with_degree(with_lwe_size(tfhe_parameters_128(), lwe_size), 4096), [NAND, XOR, MUX])
mint_mk(joinpath(outdir, "ref_mk2_n512.tfhe"), with_degree(with_lwe_size(mktfhe_parameters_2party, lwe_size), 512), 2)
