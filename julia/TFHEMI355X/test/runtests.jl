# Runs the reference's own test cases (TFHE.jl test/runtests.jl:8-100) THROUGH the GPU binding, plus the in-Julia parity check
# the Python side cannot make: the same ciphertexts through TFHE.jl's CPU gates and through GpuCloudKey must give the same
# Int32 words.  Needs Julia, a checkout of nucypher/TFHE.jl, the built library (make -C tfhe.jl_amd/csrc) and an MI355X:
#
#     julia --project=julia/TFHEMI355X -e 'using Pkg; Pkg.develop(path="<TFHE.jl checkout>"); Pkg.test()'
#
# NOT EXECUTED IN THE BUILD IMAGE (no Julia there).  tests/test_julia_shim.py checks this file statically.
using Test
using Random
using TFHE
using TFHEMI355X
using Base.Iterators: product

# test/runtests.jl:8-21 of the reference, verbatim in content
gate_tests = [
    ("NAND", gate_nand, 2, !&),
    ("OR", gate_or, 2, |),
    ("AND", gate_and, 2, &),
    ("XOR", gate_xor, 2, xor),
    ("XNOR", gate_xnor, 2, (x, y) -> xor(x, ~y)),
    ("NOT", gate_not, 1, ~),
    ("NOR", gate_nor, 2, !|),
    ("ANDNY", gate_andny, 2, (x, y) -> (~x) & y),
    ("ANDYN", gate_andyn, 2, (x, y) -> x & (~y)),
    ("ORNY", gate_orny, 2, (x, y) -> (~x) | y),
    ("ORYN", gate_oryn, 2, (x, y) -> x | (~y)),
    ("MUX", gate_mux, 3, (x, y, z) -> x ? y : z),
]

same_words(x::LweSample, y::LweSample) = x.a == y.a && x.b == y.b

@testset "TFHEMI355X" begin

    @testset "gate truth tables through GpuCloudKey (test/runtests.jl:26-40)" begin
        rng = MersenneTwister(123)
        secret_key, cloud_key = make_key_pair(rng)
        gck = GpuCloudKey(cloud_key)
        for (name, gate, nargs, reference) in gate_tests
            for bits in product([(false, true) for i in 1:nargs]...)
                ebits = [encrypt(rng, secret_key, b) for b in bits]
                eres = gate(gck, ebits...)
                @test decrypt(secret_key, eres) == reference(bits...)
                # the in-Julia parity check: the reference's CPU path on the same ciphertexts, word for word
                @test same_words(eres, gate(cloud_key, ebits...))
            end
        end

        # docs/src/manual.md:28-35: broadcasting is one GPU batch and equals the CPU broadcast word for word
        bits1 = rand(rng, Bool, 16)
        bits2 = rand(rng, Bool, 16)
        c1 = [encrypt(rng, secret_key, b) for b in bits1]
        c2 = [encrypt(rng, secret_key, b) for b in bits2]
        gpu = gate_xor.(gck, c1, c2)
        cpu = gate_xor.(cloud_key, c1, c2)
        @test all(same_words.(gpu, cpu))
        @test [decrypt(secret_key, c) for c in gpu] == xor.(bits1, bits2)
        # device-resident operands: nothing leaves the GPU between the two gates
        d1, d2 = upload(gck, c1), upload(gck, c2)
        d3 = gate_and.(gck, gate_xor.(gck, d1, d2), d1)
        @test all(same_words.(download(d3), gate_and.(cloud_key, cpu, c1)))
        # the streaming form
        t = gates_batch_async(gck, fill(TFHEMI355X.XOR, 16), c1, c2)
        @test all(same_words.(fetch(t), cpu))
        # gate_constant / gate_not are not bootstrapped (src/gates.jl:76-93)
        @test same_words(gate_constant(gck, true), gate_constant(cloud_key, true))
        @test same_words(gate_not(gck, c1[1]), gate_not(cloud_key, c1[1]))
    end

    @testset "single party, custom parameters (test/runtests.jl:43-57)" begin
        rng = MersenneTwister(123)
        params = tfhe_parameters_128()
        secret_key, cloud_key = make_key_pair(rng, params)
        gck = GpuCloudKey(cloud_key)
        for bits in product((false, true), (false, true))
            ebits = [encrypt(rng, secret_key, b) for b in bits]
            eres = gate_nand(gck, ebits...)
            @test decrypt(secret_key, eres) == !(bits[1] && bits[2])
            @test same_words(eres, gate_nand(cloud_key, ebits...))
        end
    end

    @testset "tlwe_mask_size = 2 (src/api.jl:30 keyword)" begin
        rng = MersenneTwister(123)
        secret_key, cloud_key = make_key_pair(rng, tfhe_parameters_80(tlwe_mask_size=2))
        gck = GpuCloudKey(cloud_key)
        x, y = encrypt(rng, secret_key, true), encrypt(rng, secret_key, false)
        @test same_words(gate_nand(gck, x, y), gate_nand(cloud_key, x, y))
    end

    @testset "cloud key generated on the GPU" begin
        rng = MersenneTwister(7)
        secret_key = SecretKey(rng, tfhe_parameters_80())
        gck = GpuCloudKey(rng, secret_key)
        for bits in product((false, true), (false, true))
            ebits = [encrypt(rng, secret_key, b) for b in bits]
            @test decrypt(secret_key, gate_nand(gck, ebits...)) == !(bits[1] && bits[2])
        end
    end

    @testset "multikey NAND (test/runtests.jl:60-100)" begin
        parties = 2
        params = mktfhe_parameters_2party
        rng = MersenneTwister()
        secret_keys = [SecretKey(rng, params) for i in 1:parties]
        shared_key = SharedKey(rng, params)
        ck_parts = [CloudKeyPart(rng, secret_key, shared_key) for secret_key in secret_keys]
        cloud_key = MKCloudKey(ck_parts)
        mck = GpuMKCloudKey(cloud_key)
        for trial = 1:10
            mess1 = rand(Bool)
            mess2 = rand(Bool)
            enc_mess1 = mk_encrypt(rng, secret_keys, mess1)
            enc_mess2 = mk_encrypt(rng, secret_keys, mess2)
            enc_out = mk_gate_nand(mck, enc_mess1, enc_mess2)
            ref_out = mk_gate_nand(cloud_key, enc_mess1, enc_mess2)
            # word for word first: the decrypt-level check is ~0.2 %/gate flaky in the reference itself (SURVEY §4)
            @test enc_out.a == ref_out.a && enc_out.b == ref_out.b
            @test mk_decrypt(secret_keys, enc_out) == mk_decrypt(secret_keys, ref_out)
        end
    end

    @testset "golden fixtures from the reference (scripts/mint_fixtures.jl)" begin
        outdir = mktempdir()
        # runs the script as a program would: ARGS = [outdir, lwe_size]
        empty!(ARGS); push!(ARGS, outdir, "16")
        include(joinpath(@__DIR__, "..", "scripts", "mint_fixtures.jl"))
        for f in ("ref_gates80.tfhe", "ref_gates128.tfhe", "ref_mk2.tfhe")
            @test filesize(joinpath(outdir, f)) > 1000
        end
    end
end
