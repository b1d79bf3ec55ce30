"""N > 1 host logic on CPU: contiguous shards balanced by rotation count, and the single result gather,
exercised with 2 gloo ranks.  The ranks' "engine" here is the oracle (tests may use it); on GPUs the same
code path runs with the HIP engine and RCCL (bench.py)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_cover_and_balance(tfhe):
    from tfhe_jl_amd.sharding import rotation_cost, shard_bounds
    rng = np.random.default_rng(789)
    names = ["NAND", "AND", "OR", "XOR", "MUX"]
    ops = np.array([tfhe.OPCODES[names[i]] for i in rng.integers(0, 5, 65536)], np.uint8)
    b = shard_bounds(ops, 8)
    assert b[0][0] == 0 and b[-1][1] == ops.size
    assert all(b[i][1] == b[i + 1][0] for i in range(7))
    rot = [int(rotation_cost(ops[s:e]).sum()) for s, e in b]
    assert max(rot) - min(rot) <= 4                      # balanced by rotations, not by gate count
    assert shard_bounds(np.zeros(0, np.uint8), 4) == [(0, 0)] * 4
    small = shard_bounds(np.zeros(3, np.uint8), 8)       # fewer gates than ranks: some shards empty
    assert sum(e - s for s, e in small) == 3 and all(e >= s for s, e in small)
    assert [int(c) for c in rotation_cost([tfhe.OPCODES[x] for x in ("NAND", "MUX", "NOT", "CONST1")])] == [1, 2, 0, 0]


def test_library_sharding_rule_equals_host_rule(tfhe):
    """tfhe_shard_bounds (what a multi-device context uses to split tfhe_gates_batch) == sharding.shard_bounds
    (what the one-process-per-GPU launch uses), on mixed, trivial-heavy, tiny and empty streams."""
    from tfhe_jl_amd.sharding import shard_bounds
    rng = np.random.default_rng(5)
    for B in (0, 1, 3, 7, 64, 1000, 65536):
        for world in (1, 2, 3, 8):
            ops = rng.integers(0, 15, B).astype(np.uint8)
            assert tfhe._lib.shard_bounds(ops, world) == shard_bounds(ops, world), (B, world)
    triv = np.full(100, tfhe.OPCODES["NOT"], np.uint8)
    assert tfhe._lib.shard_bounds(triv, 4) == shard_bounds(triv, 4)
    assert tfhe._lib.shard_bounds(None, 4) == [(0, 0)] * 4


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import oracle
    import tfhe_jl_amd as tfhe
    from tfhe_jl_amd.sharding import gather_shards, gather_to_root, shard_bounds
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        params = tfhe.SchemeParameters(6, 1 / 2**15, 1024, 1, 2, 10, 9e-9, 8, 2, 1 / 2**15, 1)
        rng = np.random.default_rng(123)                 # identical keys on every rank
        sk, ck = tfhe.make_key_pair(rng, params)
        o = oracle.Oracle(6, 1024, 1, 2, 10, 8, 2)
        o.load_bootstrap_key(ck.bootstrap_key)
        o.load_keyswitch_key(ck.keyswitch_key)
        names = ["NAND", "MUX", "NOT", "XOR", "MUX", "AND", "CONST1", "OR", "MUX", "NAND", "COPY"]
        ops = np.array([tfhe.OPCODES[x] for x in names], np.uint8)
        irng = np.random.default_rng(7)
        ins = [irng.integers(-2**31, 2**31, size=(ops.size, 7), dtype=np.int64).astype(np.int32) for _ in range(3)]
        bounds = shard_bounds(ops, world)
        s, e = bounds[rank]
        local = o.gates(ops[s:e], *[a[s:e] for a in ins]) if e > s else np.zeros((0, 7), np.int32)
        full = gather_shards(torch.from_numpy(local), bounds, rank).numpy()
        want = o.gates(ops, *ins)
        ok = bool(np.array_equal(full, want))
        rooted = gather_to_root(torch.from_numpy(local), bounds, rank, dst=0)      # what bench.py uses
        ok = ok and ((rooted is None) if rank != 0 else bool(np.array_equal(rooted.numpy(), want)))
        q.put((rank, ok, bounds))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_gather_matches_single_process():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    bounds = res[0][2]
    assert bounds[0][1] == bounds[1][0] and bounds[1][1] == 11
