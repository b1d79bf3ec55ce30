#!/usr/bin/env python3
"""Soak run (lives under tests/ because it uses the oracle as the checker; not collected by pytest — run it as
`python tests/soak_run.py [iterations]` on a GPU box): random batch sizes and opcode mixes against the oracle (every output word for small batches, a random
sample for large ones) plus decrypt checks, alternating the host-buffer API, the wire-table level API and — on a
{0, 0} multi-device context — streamed submits, sharded levels and CHAINS of dependent sharded levels (every level reading a
random selection of what earlier levels wrote on either device context, with the exchange path and the split threshold drawn at
random) against the same chain on one device, every wire of the table."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))                    # tests/ (conftest.KeySet)
import tfhe_jl_amd as tfhe, oracle
from conftest import KeySet
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
K = KeySet(tfhe, oracle, tfhe.tfhe_parameters_80())
eng = K.ck.engine(0)
multi = K.ck.engine([0, 0])
multi.set_option("level_split_min", 64)
rng = np.random.default_rng(2026)
names = list(tfhe.OPCODES)
truth = {"NAND": lambda x, y, z: ~(x & y), "OR": lambda x, y, z: x | y, "AND": lambda x, y, z: x & y, "XOR": lambda x, y, z: x ^ y,
         "XNOR": lambda x, y, z: ~(x ^ y), "NOT": lambda x, y, z: ~x, "NOR": lambda x, y, z: ~(x | y), "ANDNY": lambda x, y, z: ~x & y,
         "ANDYN": lambda x, y, z: x & ~y, "ORNY": lambda x, y, z: ~x | y, "ORYN": lambda x, y, z: x | ~y,
         "MUX": lambda x, y, z: np.where(x, y, z), "CONST0": lambda x, y, z: np.zeros_like(x), "CONST1": lambda x, y, z: np.ones_like(x),
         "COPY": lambda x, y, z: x}
t0 = time.time()
for it in range(iters):
    B = int(rng.choice([1, 2, 7, 9, 64, 100, 511, 513, 900, 1000, 1023, 1025, 1400, 2049, 3000]))
    sel = rng.integers(0, len(names), B)
    ops = np.array([tfhe.OPCODES[names[s]] for s in sel], np.uint8)
    bits = [rng.integers(0, 2, B).astype(bool) for _ in range(3)]
    ins = [tfhe.encrypt(K.rng, K.sk, b).data for b in bits]
    mode = ("batch", "level", "multi-submit", "multi-level", "multi-chain")[it % 5]
    if mode == "multi-chain":
        # 3 .. 6 dependent levels over a table of W wires; level t writes a fresh block of wires and reads any earlier ones
        W0 = int(rng.integers(8, 200))
        depth = int(rng.integers(3, 7))
        widths = [int(rng.integers(1, 150)) for _ in range(depth)]
        total = W0 + sum(widths)
        base = tfhe.encrypt(K.rng, K.sk, rng.integers(0, 2, W0).astype(bool)).data
        multi.set_option("level_exchange", int(rng.integers(0, 3)))
        multi.set_option("level_split_min", int(rng.choice([2, 16, 64, 4096])))
        levels, have = [], W0
        for wdt in widths:
            lops = np.array([tfhe.OPCODES[names[s_]] for s_ in rng.integers(0, len(names), wdt)], np.uint8)
            levels.append((lops, *(rng.integers(0, have, wdt).astype(np.int32) for _ in range(3)), np.arange(have, have + wdt, dtype=np.int32)))
            have += wdt
        tables = []
        for e in (eng, multi):
            e.wires_alloc(total)
            e.wires_upload(0, base)
            for lops, a_, b_, c_, o_ in levels:
                e.gates_level(lops, a_, b_, c_, o_)
            tables.append(e.wires_gather(np.arange(total, dtype=np.int32)))
        assert np.array_equal(tables[0], tables[1]), f"iter {it}: multi-device chain differs from one device"
        multi.set_option("level_exchange", 0)
        multi.set_option("level_split_min", 64)
        print(f"iter {it:3d} chain of {depth} levels, {total} wires ok", flush=True)
        continue
    if mode == "batch":
        got = eng.gates(ops, *ins)
    elif mode == "multi-submit":             # two device contexts, every one takes its shard as a submit of its own
        t, got = multi.gates_submit(ops, *ins)
        multi.gates_wait(t)
    else:                                    # same gates through the wire table, one level (sharded on the multi-device context)
        e = eng if mode == "level" else multi
        e.wires_alloc(4 * B)
        e.wires_upload(0, np.concatenate(ins))
        idx = np.arange(B, dtype=np.int32)
        e.gates_level(ops, idx, idx + B, idx + 2 * B, idx + 3 * B)
        got = e.wires_download(3 * B, B)
    want_bits = np.zeros(B, bool)
    for k, nm in enumerate(names):
        m = sel == k
        if m.any():
            want_bits[m] = truth[nm](bits[0][m], bits[1][m], bits[2][m])
    assert np.array_equal(tfhe.decrypt(K.sk, got), want_bits), f"iter {it}: decrypt mismatch"
    samp = np.arange(B) if B <= 128 else rng.choice(B, 128, replace=False)
    want = K.oracle.gates(ops[samp], *[a[samp] for a in ins], nthreads=32)
    assert np.array_equal(got[samp], want), f"iter {it}: word mismatch"
    print(f"iter {it:3d} B={B:5d} ok ({mode})", flush=True)
print(f"soak ok: {iters} iterations in {time.time() - t0:.1f} s")
