#!/usr/bin/env python3
"""bench.py — bootstrapped gates/sec on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--gates B] [--params 80|128]

One "step" = one pass of the hot path (gate prologue -> blind rotate -> extract -> keyswitch) over one
batch of B = 4096 independent NAND gates per GPU (BASELINE config 2: "batch of 4096 independent
gate_nand() bootstraps, N=1024, 1xMI355X"), inputs resident in HBM before the timed region.
N > 1: one process per GPU (torch.distributed / RCCL), every rank runs its own 4096-gate shard
(weak scaling: independent gates, replicated keys) and the shards' results are gathered with one RCCL
all_gather per step — the only collective.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BR_BYTES = {  # algorithmic key bytes per blind rotation: n*l*(k+1)^2*N*4
    "80": 500 * 2 * 4 * 1024 * 4,     # 16 384 000
    "128": 630 * 3 * 4 * 1024 * 4,    # 30 965 760
}
HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gates", type=int, default=4096, help="gates per GPU per step")
    ap.add_argument("--params", choices=["80", "128"], default="80")
    ap.add_argument("--workload", choices=["nand", "mixed"], default="nand",
                    help="nand: BASELINE config 2 per GPU (weak scaling, the default the driver runs); mixed: BASELINE config 3, "
                         "65 536 i.i.d. {NAND, AND, OR, XOR, MUX} gates in total, sharded over the GPUs by rotation count (strong scaling)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL result gather (N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-per-thread", type=int, default=8,
                    help="gates per host thread in the cpu_baseline sample (8 x 128 threads x ~30 ms = ~30 s of CPU work)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run --nproc-per-node N")
    n_gpus = world

    import torch
    import torch.distributed as dist
    import tfhe_jl_amd as tfhe

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
    # Rehearsal mode for a one-GPU box (tests/test_bench_contract.py): TFHE_BENCH_SHARE_GPU=1 puts every rank on
    # device 0 and uses gloo for the barrier / max-reduce (RCCL refuses two ranks on one device); no result gather.
    share_gpu = os.environ.get("TFHE_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
        args.no_gather = True
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if n_gpus > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    # --- keys: identical on every rank (seed 123), replicated per GPU ------------------------------
    params = tfhe.tfhe_parameters_80() if args.params == "80" else tfhe.tfhe_parameters_128()
    krng = np.random.default_rng(123)
    sk, ck = tfhe.make_key_pair(krng, params)
    eng = ck.engine(local_rank)

    # --- inputs: encryptions of i.i.d. uniform bits, seed 456 (+rank), uploaded before timing -------
    n1 = params.lwe_size + 1
    if args.workload == "nand":
        B = args.gates
        irng = np.random.default_rng(456 + rank)
        bx, by = irng.integers(0, 2, B).astype(bool), irng.integers(0, 2, B).astype(bool)
        bz = np.zeros(B, bool)
        ops = np.zeros(B, np.uint8)  # NAND
        expect = ~(bx & by)
        total_per_step = world * B
    else:
        from tfhe_jl_amd.sharding import shard_bounds
        mrng = np.random.default_rng(789)                       # identical stream on every rank, then sliced
        BT = 65536
        names = ["NAND", "AND", "OR", "XOR", "MUX"]
        sel = mrng.integers(0, 5, BT)
        all_ops = np.array([tfhe.OPCODES[x] for x in names], np.uint8)[sel]
        bits = [mrng.integers(0, 2, BT).astype(bool) for _ in range(3)]
        s0, e0 = shard_bounds(all_ops, world)[rank]
        ops, sel = all_ops[s0:e0], sel[s0:e0]
        bx, by, bz = (b[s0:e0] for b in bits)
        B = e0 - s0
        expect = np.select([sel == 0, sel == 1, sel == 2, sel == 3, sel == 4], [~(bx & by), bx & by, bx | by, bx ^ by, np.where(bx, by, bz)])
        total_per_step = BT
        irng = np.random.default_rng(456 + rank)
    hx, hy, hz = (tfhe.encrypt(irng, sk, b).data for b in (bx, by, bz))
    dx, dy, dz = (torch.from_numpy(h).to(dev) for h in (hx, hy, hz))
    dout = torch.empty((B, n1), dtype=torch.int32, device=dev)
    gathered = torch.empty((world * B, n1), dtype=torch.int32, device=dev) if (n_gpus > 1 and args.workload == "nand") else None
    stream = torch.cuda.current_stream(dev).cuda_stream

    br_ms, ks_ms = [], []

    def step(record):
        eng.gates_dev(ops, dx.data_ptr(), dy.data_ptr(), dz.data_ptr(), dout.data_ptr(), B, stream)
        if gathered is not None and not args.no_gather:
            dist.all_gather_into_tensor(gathered, dout)
        if record:
            br_ms.append(eng.last_timing_ms(0))   # HIP events on `stream` around the blind-rotate kernel
            ks_ms.append(eng.last_timing_ms(1))

    def barrier():
        if n_gpus > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    rotations_per_step = eng.last_rotation_count()   # of the timed launches (read before any other call on eng)
    if n_gpus > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # --- "ms/bootstrap" half of the metric: latency of ONE gate_nand (B = 1), outside the timed region -------
    single_ms = None
    if rank == 0 and args.workload == "nand":
        one = np.zeros(1, np.uint8)
        o1 = torch.empty((1, n1), dtype=torch.int32, device=dev)
        lat = []
        for it in range(25):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            eng.gates_dev(one, dx.data_ptr(), dy.data_ptr(), dz.data_ptr(), o1.data_ptr(), 1, stream)
            torch.cuda.synchronize(dev)
            lat.append(time.perf_counter() - t1)
        single_ms = float(np.median(lat[5:])) * 1e3
        assert torch.equal(o1[0], dout[0]), "single-gate result differs from the batched one"

    # --- correctness of what was just timed (not timed itself) ---------------------------------------
    out = dout.cpu().numpy()
    ok_decrypt = bool(np.array_equal(tfhe.decrypt(sk, out), expect))

    result = None
    if rank == 0:
        br_avg_s = float(np.mean(br_ms)) * 1e-3
        achieved = rotations_per_step * BR_BYTES[args.params] / br_avg_s
        total_gates = total_per_step * args.steps
        result = {
            "metric": "bootstrapped gates/sec (whole node), N=1024",
            "value": total_gates / elapsed,
            "unit": "gates/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_bootstrap_amortised": elapsed / args.steps * 1e3 / B,
            "ms_per_bootstrap_single_gate": single_ms,
            "higher_is_better": True,
            "scaling": "weak" if args.workload == "nand" else "strong",
            "vs_baseline": None,
            "dtype": "int32 torus / f64 transform",
            "data": "synthetic: oracle-independent numpy keygen (seed 123), encryptions of uniform random bits (seed 456)",
            "config": {
                "workload": (f"batch of {B} independent gate_nand() bootstraps per GPU" if args.workload == "nand" else
                             f"mixed gate stream (NAND/AND/OR/XOR/MUX), 65536 gates in total, {B} on rank 0") + f", tfhe_parameters_{args.params} "
                            f"(n={params.lwe_size}, N=1024, k=1, l={params.bs_decomp_length}, Bg=2^{params.bs_log2_base}, "
                            f"ks t={params.ks_decomp_length}, base 2^{params.ks_log2_base})",
                "gates_per_gpu_per_step": B,
                "result_gather": "rccl all_gather" if (gathered is not None and not args.no_gather) else "none",
            },
            "outputs_decrypt_correctly": ok_decrypt,
            "roofline": {
                "bound": "hbm",
                "kernel": "blind_rotate_kernel_v3",
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "traffic": traffic_from_profiles("blind_rotate_kernel_v3", rotations_per_step),
                "bytes_per_unit": BR_BYTES[args.params],
                "units_per_launch": rotations_per_step,
                "avg_launch_ms": br_avg_s * 1e3,
                "keyswitch_avg_launch_ms": float(np.mean(ks_ms)),
            },
        }
        if n_gpus == 1 and not args.no_cpu_baseline and args.workload == "nand":
            result["cpu_baseline"] = cpu_baseline(tfhe, params, ck, hx, hy, out, args)
        print(json.dumps(result), flush=True)
    if n_gpus > 1:
        dist.barrier()
        dist.destroy_process_group()
    ck.close()
    if not ok_decrypt:
        sys.exit("bench.py: GPU outputs did not decrypt to the gates' truth values")


def traffic_from_profiles(kernel_substr, units_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/<tag>/traffic.json, written by tools/prof_summary.py: FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024,
    separate --pmc passes of this same command).  Only reported when that profile was taken at the same
    number of rotations per launch; otherwise null."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        meta = d.get("_meta", {})
        if meta.get("units_per_launch") not in (None, units_per_launch):
            continue
        for k, v in d.items():
            if kernel_substr in k:
                best = v["hbm_bytes_per_launch"]
    return best


def cpu_baseline(tfhe, params, ck, hx, hy, gpu_out, args):
    """The oracle (C restatement of the reference algorithm, reference-style Float64 FFT) timed on this
    box's host cores on a bounded sample of the same workload; also re-checks parity on that sample."""
    import oracle
    oracle.build()
    p = params
    o = oracle.Oracle(p.lwe_size, p.tlwe_polynomial_degree, p.tlwe_mask_size, p.bs_decomp_length, p.bs_log2_base,
                      p.ks_decomp_length, p.ks_log2_base)
    o.load_bootstrap_key(ck.bootstrap_key)
    o.load_keyswitch_key(ck.keyswitch_key)
    threads = oracle.max_threads()
    S = min(hx.shape[0], max(64, args.cpu_sample_per_thread * threads))
    ops = np.zeros(S, np.uint8)
    t0 = time.perf_counter()
    want = o.gates(ops, hx[:S], hy[:S], nthreads=threads)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    o.gates(ops[:4], hx[:4], hy[:4], nthreads=1)
    dt1 = (time.perf_counter() - t1) / 4
    return {
        "value": S / dt,
        "unit": "gates/s",
        "cores": threads,
        "kind": "port",
        "sample": f"first {S} of the {hx.shape[0]} NAND gates of the GPU workload, one gate per OpenMP thread; "
                  f"C restatement of the reference algorithm (not Julia)",
        "single_thread_ms_per_gate": dt1 * 1e3,
        "parity_on_sample": bool(np.array_equal(want, gpu_out[:S])),
    }


if __name__ == "__main__":
    main()
