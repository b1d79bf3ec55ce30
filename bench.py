#!/usr/bin/env python3
"""bench.py — bootstrapped gates/sec on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--gates B] [--params 80|128] [--workload nand|mixed]

One "step" = one pass of the hot path (gate prologue -> blind rotate -> extract -> keyswitch) over one
batch of B = 4096 independent NAND gates per GPU (BASELINE config 2: "batch of 4096 independent
gate_nand() bootstraps, N=1024, 1xMI355X"), inputs resident in HBM before the timed region.

N > 1, either launch style gives the same one-process-per-GPU job (torch.distributed over RCCL):
  * under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (the driver's line), or
  * plain `python bench.py --gpus N`: this process then only starts N fresh rank processes itself — before it
    imports torch or touches a GPU — relays rank 0's JSON line and exits non-zero if any rank failed.
Every rank runs its own shard (nand: 4096 gates per GPU, weak scaling; mixed: BASELINE config 3, 65 536 gates
in total cut into rotation-balanced shards, strong scaling) with no communication during compute; the shards'
results are gathered to rank 0 with ONE RCCL gather per step — the only collective.  Rank 0 prints ONE JSON line.

`--fanout` is the other way to use N GPUs: ONE process, ONE multi-device context (tfhe_ctx_create_multi) whose
tfhe_gates_batch splits the host batch over the GPUs inside the library (what a Julia caller gets).
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK = 78.6e12   # FLOP/s vector FP64 at the 2.4 GHz maximum clock (256 CUs x 4 SIMDs x 32 FLOP/clk)
MAX_CLOCK_MHZ = 2400.0


def br_bytes(p):
    """Algorithmic key bytes per blind rotation, n*l*(k+1)^2*N*4 (canonical Int32 key, no reuse credited)."""
    return p.lwe_size * p.bs_decomp_length * (p.tlwe_mask_size + 1) ** 2 * p.tlwe_polynomial_degree * 4


def br_flops(p):
    """Algorithmic FP64 flops per blind rotation (SURVEY §8d "secondary"): per CMUX step (k+1)(l+1) transforms of
    5 M log2 M flops (M = N/2) plus (k+1)^2 l M complex multiply-adds of 8 flops; n steps."""
    M = p.tlwe_polynomial_degree // 2
    k1, l = p.tlwe_mask_size + 1, p.bs_decomp_length
    return p.lwe_size * (k1 * (l + 1) * 5 * M * int(np.log2(M)) + k1 * k1 * l * M * 8)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--gates", type=int, default=4096, help="gates per GPU per step")
    ap.add_argument("--params", choices=["80", "128"], default="80")
    ap.add_argument("--workload", choices=["nand", "mixed"], default="nand",
                    help="nand: BASELINE config 2 per GPU (weak scaling, the default the driver runs); mixed: BASELINE config 3, "
                         "65 536 i.i.d. {NAND, AND, OR, XOR, MUX} gates in total, sharded over the GPUs by rotation count (strong scaling)")
    ap.add_argument("--no-gather", action="store_true", help="skip the RCCL result gather (N > 1)")
    ap.add_argument("--fanout", action="store_true",
                    help="one process, one multi-device context: the library splits a host batch over --gpus devices itself")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the in-kernel clock / rounding-margin run and the PCIe-inclusive loop")
    ap.add_argument("--cpu-sample-per-thread", type=int, default=8,
                    help="gates per host thread in the cpu_baseline sample (8 x 128 threads x ~30 ms = ~30 s of CPU work)")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves.  Runs before torch or the
    HIP library is imported, so this parent never touches a GPU; children are fresh processes (no exec of a process
    that has initialised the GPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.exit(f"bench.py: rank(s) failed: {bad}")
    sys.exit(0)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None and not args.fanout:
        spawn_ranks(args.gpus)          # never returns

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.fanout:
        if world != 1:
            sys.exit("bench.py --fanout is a single-process mode")
    elif args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    n_gpus = args.gpus

    import torch
    import torch.distributed as dist
    import tfhe_jl_amd as tfhe

    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the engine has no CPU fallback)")
    # Rehearsal mode for a one-GPU box (tests/test_bench_contract.py): TFHE_BENCH_SHARE_GPU=1 puts every rank (or every
    # device context of --fanout) on device 0 and uses gloo for the barrier / max-reduce / gather (RCCL refuses two
    # ranks on one device).
    share_gpu = os.environ.get("TFHE_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # One explicit stream for everything: the engine's kernels are launched on it (a NULL stream would mean the context's
    # own private stream), and torch's copies and the RCCL gather order themselves against torch's CURRENT stream.
    work_stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(work_stream)
    # TFHE_BENCH_FORCE_DIST=1: a single rank still goes through the process group (RCCL init, barrier, max-reduce and the
    # gather with one participant) — the only way to execute those calls on a one-GPU box (tests/test_bench_contract.py)
    use_dist = (world > 1 or os.environ.get("TFHE_BENCH_FORCE_DIST") == "1") and not args.fanout
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(so.getsockname()[1])
        tmo = datetime.timedelta(seconds=300)          # a rank that died must not leave the others waiting for half an hour
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)

    # --- keys: identical on every rank (seed 123), replicated per GPU ------------------------------
    params = tfhe.tfhe_parameters_80() if args.params == "80" else tfhe.tfhe_parameters_128()
    krng = np.random.default_rng(123)
    sk, ck = tfhe.make_key_pair(krng, params)
    if args.fanout:
        eng = ck.engine([0] * n_gpus if share_gpu else list(range(n_gpus)))
    else:
        eng = ck.engine(local_rank)

    # --- inputs: encryptions of i.i.d. uniform bits, seed 456 (+rank), uploaded before timing -------
    n1 = params.lwe_size + 1
    from tfhe_jl_amd.sharding import shard_bounds
    parts = 1 if args.fanout else world          # shards handled by separate processes
    if args.workload == "nand":
        B = args.gates * (n_gpus if args.fanout else 1)
        irng = np.random.default_rng(456 + rank)
        bx, by = irng.integers(0, 2, B).astype(bool), irng.integers(0, 2, B).astype(bool)
        bz = np.zeros(B, bool)
        ops = np.zeros(B, np.uint8)  # NAND
        expect = ~(bx & by)
        total_per_step = parts * B
        bounds = [(r * B, (r + 1) * B) for r in range(parts)]
    else:
        mrng = np.random.default_rng(789)                       # identical stream on every rank, then sliced
        BT = 65536
        names = ["NAND", "AND", "OR", "XOR", "MUX"]
        sel = mrng.integers(0, 5, BT)
        all_ops = np.array([tfhe.OPCODES[x] for x in names], np.uint8)[sel]
        bits = [mrng.integers(0, 2, BT).astype(bool) for _ in range(3)]
        bounds = shard_bounds(all_ops, parts)
        s0, e0 = bounds[rank]
        ops, sel = all_ops[s0:e0], sel[s0:e0]
        bx, by, bz = (b[s0:e0] for b in bits)
        B = e0 - s0
        expect = np.select([sel == 0, sel == 1, sel == 2, sel == 3, sel == 4], [~(bx & by), bx & by, bx | by, bx ^ by, np.where(bx, by, bz)])
        total_per_step = BT
        irng = np.random.default_rng(456 + rank)
    hx, hy, hz = (tfhe.encrypt(irng, sk, b).data for b in (bx, by, bz))
    use_gather = use_dist and not args.no_gather
    if not args.fanout:
        dx, dy, dz = (torch.from_numpy(h).to(dev) for h in (hx, hy, hz))
        # two result buffers: the gather of step i (a point-to-point transfer to rank 0 over xGMI) overlaps step i + 1
        douts = [torch.empty((B, n1), dtype=torch.int32, device=dev) for _ in range(2)]
        dout = douts[0]
        stream = work_stream.cuda_stream
        assert stream, "the engine must be given torch's current stream explicitly"
    host_out = None
    br_ms, ks_ms = [], []

    class RootGather:
        """One gather of the shards' results to rank 0 per step, asynchronous: everything is allocated up front, shards
        are padded to the longest one, rank 0 keeps the received parts.  Transport: RCCL (`dist.gather` = grouped
        send/recv to the root); on a one-GPU rehearsal gloo on host copies.  If this build's RCCL backend rejects
        `gather`, the plain all_gather (every rank receives everything) is used instead and the JSON line says so."""

        def __init__(self):
            self.longest = max(e - s for s, e in bounds)
            self.mode = "gather"
            tdev = "cpu" if share_gpu else dev
            self.padded = [torch.zeros((self.longest, n1), dtype=torch.int32, device=tdev) for _ in range(2)]
            self.parts = [[torch.empty((self.longest, n1), dtype=torch.int32, device=tdev) for _ in range(world)] if rank == 0 else None
                          for _ in range(2)]
            self.full = None
            self.pending = [None, None]
            # one complete gather now (untimed): an asynchronous collective reports a backend problem at wait(), not at the
            # call, so find out before the timed loop which transport works
            for mode in ("gather", "all_gather"):
                self.mode = mode
                try:
                    self.launch(0, self.padded[1][:B])
                    self.wait(0)
                    if not share_gpu:
                        torch.cuda.synchronize(dev)
                    break
                except (RuntimeError, NotImplementedError) as e:
                    print(f"bench.py: rank {rank}: {mode} to rank 0 failed ({e})", file=sys.stderr)
                    self.pending[0] = None
            else:
                sys.exit("bench.py: neither gather nor all_gather works on this process group")

        def wait(self, k):
            if self.pending[k] is not None:
                self.pending[k].wait()
                self.pending[k] = None

        def launch(self, k, src):
            if share_gpu:
                self.padded[k][:B].copy_(src)                 # D2H (synchronous): rehearsal only
                send = self.padded[k]
            elif B == self.longest:
                send = src
            else:
                self.padded[k][:B].copy_(src)
                send = self.padded[k]
            if self.mode == "gather":
                self.pending[k] = dist.gather(send, self.parts[k], dst=0, async_op=True)
                return
            if self.full is None:
                self.full = [torch.empty((world * self.longest, n1), dtype=torch.int32, device=send.device) for _ in range(2)]
            self.pending[k] = dist.all_gather_into_tensor(self.full[k], send, async_op=True)

        def result(self, k):
            """Rank 0: the gathered [total][n+1] result of buffer k (after wait)."""
            if rank != 0:
                return None
            if self.mode == "gather":
                return torch.cat([self.parts[k][r][: e - s] for r, (s, e) in enumerate(bounds)], dim=0)
            return torch.cat([self.full[k][r * self.longest: r * self.longest + (e - s)] for r, (s, e) in enumerate(bounds)], dim=0)

    gatherer = RootGather() if (use_gather and not args.fanout) else None
    step_no = 0

    if args.fanout:
        # page-locked operands and results (tfhe_host_alloc): every device's copies are single DMA transfers, and the streaming
        # entry points keep two batches in flight per device, so uploads and downloads run under the other batch's kernels
        pins = [tfhe.pinned_empty(h.shape) for h in (hx, hy, hz)]
        for p_, h in zip(pins, (hx, hy, hz)):
            p_[:] = h
        fan_outs = [tfhe.pinned_empty((B, n1)) for _ in range(2)]
        fan_tickets = [None, None]

    def step(record):
        nonlocal host_out, dout, step_no
        if args.fanout:
            k = step_no & 1
            step_no += 1
            if fan_tickets[k] is not None:
                eng.gates_wait(fan_tickets[k])
            fan_tickets[k], host_out = eng.gates_submit(ops, pins[0], pins[1], pins[2] if args.workload == "mixed" else None, out=fan_outs[k])
        else:
            k = step_no & 1
            step_no += 1
            dout = douts[k]
            if gatherer:
                gatherer.wait(k)          # the gather that read this buffer two steps ago
            eng.gates_dev(ops, dx.data_ptr(), dy.data_ptr(), dz.data_ptr(), dout.data_ptr(), B, stream)
            if gatherer:
                gatherer.launch(k, dout)

    def drain():
        if gatherer:
            gatherer.wait(0)
            gatherer.wait(1)
        if args.fanout:
            for k in (0, 1):
                if fan_tickets[k] is not None:
                    eng.gates_wait(fan_tickets[k])
                    fan_tickets[k] = None

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    drain()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    launch_ms_source = "HIP events around the kernels of the timed steps (tfhe_timing_history_ms)"
    if args.fanout:
        # kernel times of blocking calls AFTER the timed region (a multi-device context reports the slowest shard of its last
        # blocking call; the streamed calls above leave per-device histories only): the median of five, and the JSON line says so —
        # these launches take the blocking path (two halves on two streams per device from pipeline_min gates up), not the
        # streamed submits that `value` was measured on
        samples = []
        for _ in range(5):
            eng.gates(ops, pins[0], pins[1], pins[2] if args.workload == "mixed" else None, out=fan_outs[0])
            samples.append((eng.last_timing_ms(0), eng.last_timing_ms(1)))
        br_ms, ks_ms = [float(np.median([a for a, _ in samples]))], [float(np.median([b for _, b in samples]))]
        launch_ms_source = "median of 5 blocking tfhe_gates_batch calls after the timed region (slowest shard each); the timed steps were streamed submits"
        host_out = fan_outs[0]
    if not args.fanout:
        # HIP events the engine recorded around its kernels on the stream they were launched on, for the timed steps, read
        # now in one go: no synchronisation between the steps themselves (the engine keeps the events of its last 32 calls)
        br_ms = eng.timing_history_ms(0, min(args.steps, 32))
        ks_ms = eng.timing_history_ms(1, min(args.steps, 32))
    rotations_per_step = eng.last_rotation_count()   # of the timed launches (read before any other call on eng)
    kernel_name = eng.last_kernel_name()
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # --- correctness of what was just timed (not timed itself) ---------------------------------------
    out = host_out if args.fanout else dout.cpu().numpy()
    ok_decrypt = bool(np.array_equal(tfhe.decrypt(sk, out), expect))
    # The first multi-GPU run validates itself: every rank reports a position-weighted checksum of the words it computed and
    # whether they decrypt; rank 0 recomputes the checksum of every shard AS IT ARRIVED through the gather.
    def checksum(a):
        a = np.ascontiguousarray(a, np.int32).view(np.uint32).astype(np.uint64).reshape(-1)
        return int((a * (np.arange(a.size, dtype=np.uint64) % np.uint64(65521) + np.uint64(1))).sum() & np.uint64(0x7FFFFFFFFFFFFFFF))
    gather_ok = gather_sums_ok = all_ranks_decrypt = None
    if use_dist:
        mine = torch.tensor([checksum(out), int(ok_decrypt)], dtype=torch.int64, device="cpu" if share_gpu else dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        all_ranks_decrypt = bool(all(int(t[1]) == 1 for t in every))
    if gatherer and rank == 0:
        g = gatherer.result((step_no - 1) & 1).cpu().numpy()
        s0, e0 = bounds[0]
        gather_ok = bool(g.shape[0] == bounds[-1][1] and np.array_equal(g[s0:e0], out))
        gather_sums_ok = bool(g.shape[0] == bounds[-1][1] and all(checksum(g[s:e]) == int(every[r][0]) for r, (s, e) in enumerate(bounds)))
    fanout_ok = None
    if args.fanout and n_gpus > 1:
        # one device context recomputes the tail of the LAST shard (computed by the last device of the fan-out): same words
        from tfhe_jl_amd.sharding import shard_bounds as sb
        last0 = sb(ops, n_gpus)[-1][0]
        lo = max(last0, B - 256)
        one = ck.engine(0)
        ref = one.gates(ops[lo:], hx[lo:], hy[lo:], hz[lo:] if args.workload == "mixed" else None)
        fanout_ok = bool(np.array_equal(ref, out[lo:]))

    # --- extras on rank 0, outside the timed region ---------------------------------------------------
    single_ms = single_ms_noev = pcie_value = pcie_sync = pcie_pageable = clock_mhz = margin = None
    if rank == 0 and not args.fanout:
        if args.workload == "nand":   # "ms/bootstrap" half of the metric: latency of ONE gate_nand (B = 1)
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
            # the same with the engine's per-phase timing events off (option "timing_events" = 0, what circuit.py runs its levels with:
            # each of the four event records of a call costs the stream ~5 us between two kernels); reported beside, never instead
            eng.set_option("timing_events", 0)
            try:
                lat = []
                for it in range(25):
                    torch.cuda.synchronize(dev)
                    t1 = time.perf_counter()
                    eng.gates_dev(one, dx.data_ptr(), dy.data_ptr(), dz.data_ptr(), o1.data_ptr(), 1, stream)
                    torch.cuda.synchronize(dev)
                    lat.append(time.perf_counter() - t1)
                single_ms_noev = float(np.median(lat[5:])) * 1e3
                assert torch.equal(o1[0], dout[0]), "single-gate result differs from the batched one"
            finally:
                eng.set_option("timing_events", 1)
        if not args.no_diagnostics and world == 1:
            # the same step through HOST buffers (tfhe_gates_batch: 2-3 uploads + 1 download over PCIe per step)
            reps = max(2, min(args.steps, 5))
            hz_ = hz if args.workload == "mixed" else None       # NAND reads no third operand: nothing to upload for it
            eng.gates(ops, hx, hy, hz_)
            t1 = time.perf_counter()
            for _ in range(reps):
                eng.gates(ops, hx, hy, hz_)
            pcie_pageable = B * reps / (time.perf_counter() - t1)
            # ... and from / into page-locked buffers (tfhe_host_alloc), the result array reused: what a caller that keeps
            # its ciphertext arrays in such memory gets (the Julia shim's flatten buffers, a server's I/O buffers)
            px, py, pout = (tfhe.pinned_empty(hx.shape) for _ in range(3))
            px[:], py[:] = hx, hy
            pz = None
            if hz_ is not None:
                pz = tfhe.pinned_empty(hz.shape); pz[:] = hz
            eng.gates(ops, px, py, pz, out=pout)
            assert np.array_equal(pout, out), "host-buffer path differs from the device-buffer path"
            t1 = time.perf_counter()
            for _ in range(reps):
                eng.gates(ops, px, py, pz, out=pout)
            pcie_sync = B * reps / (time.perf_counter() - t1)
            # ... and STREAMED: tfhe_gates_batch_submit / _wait, two batches in flight on two streams (the upload of one batch
            # under the kernels of the other), as many steps as the timed loop above; every step's H2D + kernels + D2H is
            # inside the timed region, results checked afterwards
            sreps = max(4, args.steps)
            pouts = [pout, tfhe.pinned_empty(hx.shape)]
            tk, _ = eng.gates_submit(ops, px, py, pz, out=pouts[0]); eng.gates_wait(tk)        # (second stream's context made here)
            tk, _ = eng.gates_submit(ops, px, py, pz, out=pouts[1]); eng.gates_wait(tk)
            for o in pouts:
                assert np.array_equal(o, out), "streamed host-buffer path differs from the device-buffer path"
                o[:] = 0
            t1 = time.perf_counter()
            prev = None
            for it in range(sreps):
                tk, _ = eng.gates_submit(ops, px, py, pz, out=pouts[it & 1])
                if prev is not None:
                    eng.gates_wait(prev)
                prev = tk
            eng.gates_wait(prev)
            pcie_value = B * sreps / (time.perf_counter() - t1)
            for o in pouts:
                assert np.array_equal(o, out), "streamed host-buffer path differs from the device-buffer path"
            # clock the blind-rotate kernel holds under this load + its rounding margin: DIAG instantiation of the
            # same kernel, >= 2 s of back-to-back launches first (MI355X_MICROARCH.md, DVFS give-back item 6)
            eng.set_option("measure_margin", 1)
            t1 = time.perf_counter()
            while time.perf_counter() - t1 < 2.0:
                for _ in range(8):
                    eng.gates_dev(ops, dx.data_ptr(), dy.data_ptr(), dz.data_ptr(), dout.data_ptr(), B, stream)
                torch.cuda.synchronize(dev)
            clock_mhz = eng.last_kernel_clock_mhz()
            margin = eng.last_rounding_margin()
            eng.set_option("measure_margin", 0)

    result = None
    if rank == 0:
        br_avg_s = float(np.mean(br_ms)) * 1e-3
        achieved = rotations_per_step * br_bytes(params) / br_avg_s
        flops = rotations_per_step * br_flops(params) / br_avg_s
        total_gates = total_per_step * args.steps
        prof = profile_counters(kernel_name, rotations_per_step)
        p = params
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
            "ms_per_bootstrap_single_gate_without_timing_events": single_ms_noev,
            "higher_is_better": True,
            "scaling": "weak" if args.workload == "nand" else "strong",
            "vs_baseline": None,
            "dtype": "int32 torus / f64 transform",
            "data": "synthetic: oracle-independent numpy keygen (seed 123), encryptions of uniform random bits (seed 456)",
            "config": {
                "workload": (f"batch of {args.gates} independent gate_nand() bootstraps per GPU" if args.workload == "nand" else
                             f"mixed gate stream (NAND/AND/OR/XOR/MUX), 65536 gates in total, {B} on rank 0") + f", tfhe_parameters_{args.params} "
                            f"(n={p.lwe_size}, N=1024, k=1, l={p.bs_decomp_length}, Bg=2^{p.bs_log2_base}, "
                            f"ks t={p.ks_decomp_length}, base 2^{p.ks_log2_base})",
                "gates_per_gpu_per_step": args.gates if args.workload == "nand" else B,
                "inputs": "host buffers (PCIe inside the timed region)" if args.fanout else "resident in HBM",
                "launch": ("one process, multi-device context (tfhe_ctx_create_multi)" if args.fanout else
                           "one process per GPU (torch.distributed)" if use_dist else "one process"),
                "result_gather": ("none (results written into the caller's host buffer)" if args.fanout else
                                  (("gloo " if share_gpu else "rccl ") + (gatherer.mode if gatherer else "gather") + " to rank 0, overlapped with the next step"
                                   + (" (one-GPU rehearsal)" if share_gpu else "")) if use_gather else "none"),
            },
            "outputs_decrypt_correctly": ok_decrypt,
            "gather_matches_local_shard": gather_ok,
            "gather_matches_every_ranks_checksum": gather_sums_ok,
            "every_rank_decrypts": all_ranks_decrypt,
            "fanout_matches_one_device": fanout_ok,
            "value_pcie_inclusive": pcie_value,
            "value_pcie_inclusive_note": "the same steps streamed through tfhe_gates_batch_submit / _wait (two batches in flight, operands and "
                                         "results in page-locked host memory from tfhe_host_alloc): every step's H2D + kernels + D2H inside "
                                         "the timed region; _synchronous = one blocking tfhe_gates_batch per step (two half-batches on two "
                                         "streams inside the call), _pageable = the same from ordinary numpy arrays",
            "value_h2d_inclusive": pcie_value,      # SURVEY §8(d) config 2 asks for the host-to-device copy inside the timed region: the streamed figure has it (and the download too)
            "value_pcie_inclusive_synchronous": pcie_sync,
            "value_pcie_inclusive_pageable": pcie_pageable,
            "roofline": {
                "bound": "hbm",
                "bound_note": "judged figure (SURVEY §8d): algorithmic key bytes / launch time against HBM peak; the transformed key "
                              "(32.8 MB) is shared by all resident rotations and is served from L1 / L2 / Infinity Cache, so real HBM "
                              "traffic is ~1.5 % of this; the kernel is bound by instruction issue across FP64 VALU (floor: 64 % of the "
                              "launch with everything else removed), the LDS transpositions and the key reads at two waves per SIMD "
                              "(DESIGN.md 4.1, ablation table): see roofline_secondary",
                "kernel": kernel_name,
                "achieved": achieved / 1e9,
                "peak": HBM_PEAK / 1e9,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK,
                "frac_of_measured_copy_rate": achieved / 6.29e12,      # 6.29 TB/s float4 copy (MI355X_MICROARCH.md), quoted beside the 8 TB/s spec
                # PMC counters cannot be read from inside a run: `traffic` is REPLAYED from the committed rocprofv3 profile named in
                # traffic_replayed_from (same kernel sources by hash, same launch shape), or null — see profile_counters()
                "traffic": prof.get("hbm_bytes_per_launch"),
                "traffic_is_replayed": prof.get("hbm_bytes_per_launch") is not None,
                "traffic_replayed_from": prof.get("source"),
                "counters_note": prof.get("note"),
                "bytes_per_unit": br_bytes(params),
                "units_per_launch": rotations_per_step,
                "avg_launch_ms": br_avg_s * 1e3,
                "launch_ms_source": launch_ms_source,
                "keyswitch_avg_launch_ms": float(np.mean(ks_ms)),
            },
            "roofline_secondary": {
                "bound": "fp64_valu",
                "kernel": kernel_name,
                "flops_per_unit": br_flops(params),
                "flops_formula": "n * ((k+1)(l+1) * 5 M log2 M + (k+1)^2 l M * 8), M = N/2 (SURVEY §8d)",
                "achieved": flops / 1e12,
                "peak": FP64_VALU_PEAK / 1e12,
                "unit": "TFLOP/s",
                "frac": flops / FP64_VALU_PEAK,
                "clock_mhz": clock_mhz,
                "clock_source": "s_memtime / s_memrealtime inside the DIAG instantiation of the same kernel after >= 2 s of back-to-back launches" if clock_mhz else None,
                "frac_at_measured_clock": (flops / (FP64_VALU_PEAK * clock_mhz / MAX_CLOCK_MHZ)) if clock_mhz else None,
                "valu_busy_frac_replayed": prof.get("valu_busy_frac"),
                "valu_insts_per_launch_replayed": prof.get("valu_insts_per_launch"),
                "counters_replayed_from": prof.get("source"),
                "rounding_margin": margin,
            },
        }
        if n_gpus == 1 and not args.no_cpu_baseline and args.workload == "nand" and not args.fanout:
            result["cpu_baseline"] = cpu_baseline(tfhe, params, ck, hx, hy, out, args)
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ck.close()
    if not ok_decrypt:
        sys.exit("bench.py: GPU outputs did not decrypt to the gates' truth values")
    if gather_ok is False or gather_sums_ok is False:
        sys.exit("bench.py: the gathered result differs from what the ranks computed")
    if all_ranks_decrypt is False:
        sys.exit("bench.py: some rank's GPU outputs did not decrypt to the gates' truth values")
    if fanout_ok is False:
        sys.exit("bench.py: the multi-device context's result differs from a one-device context's")


def rocprof_symbol_prefix(kernel_name):
    """The engine's name for the launched kernel -> the start of its demangled symbol as rocprofv3 prints it:
    "blind_rotate_kernel_v3<2,8,tw2reg,rw4>" is blind_rotate_kernel_v3<2, 8, true, false, 4>, "...v3<2,8,tw2reg>" is
    <2, 8, true, false, 1>, "...v3<2,16>" is <2, 16, false, false, 1> (template arguments: l, key values prefetched, pass-B
    twiddles in registers, diagnostics, rotations per workgroup)."""
    m = re.fullmatch(r"blind_rotate_kernel_v3<(\d+),(\d+)(,tw2reg)?(,rw4)?>", kernel_name)
    if m:
        return f"blind_rotate_kernel_v3<{m.group(1)}, {m.group(2)}, {'true' if m.group(3) else 'false'}, false, {4 if m.group(4) else 1}>"
    return kernel_name.split("(")[0].rstrip(">").replace(",", ", ")


def profile_counters(kernel_name, units_per_launch):
    """PMC-derived figures for the dominant kernel from a committed rocprofv3 profile of THIS command AND THIS CODE
    (profiles/<tag>/counters.json, written by tools/prof_summary.py from separate --pmc passes): HBM bytes per launch
    (FETCH_SIZE x 1024 x 2 on gfx950 + WRITE_SIZE x 1024), VALU instructions, VALU-busy fraction.  PMC counters cannot be
    read from inside a run, so they are replayed — but only from a profile stamped with the hash of the kernel sources
    the loaded library was built from (tools/source_hash.py) and taken at the same number of rotations per launch;
    otherwise the fields are null and `counters_note` says why."""
    import glob
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import kernel_source_sha16
    want_hash = kernel_source_sha16(ROOT)
    key = rocprof_symbol_prefix(kernel_name)
    best = {}
    stale = 0
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "counters.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        meta = d.get("_meta", {})
        if "bench.py" not in meta.get("command", "bench.py"):
            continue
        if meta.get("kernel_source_sha16") != want_hash:
            stale += 1
            continue
        for k, v in d.items():
            # (the profile must be of the same launch shape: one workgroup per rotation, or per four in the ",rw4" kernels)
            per_wg = 4 if kernel_name.endswith(",rw4>") else 2 if kernel_name.endswith(",rw2>") else 1
            if k == "_meta" or v.get("workgroups_per_launch") not in (None, (units_per_launch + per_wg - 1) // per_wg):
                continue
            if re.search(r"(?<![A-Za-z0-9_])" + re.escape(key), k):
                best = dict(v)
                best["source"] = os.path.relpath(f, ROOT) + (f" ({meta.get('date')}, kernel sources {want_hash})" if meta.get("date") else "")
    if not best:
        best["note"] = (f"no committed profile of this code (kernel sources {want_hash}; {stale} profile(s) of other versions ignored): "
                        "run tools/profile.sh and commit its counters.json")
    return best


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def julia_reference_probe():
    """if this box happens to have Julia with TFHE.jl installed, time the reference's own gate_nand beside
    the port; otherwise say what is missing (the reference is Julia and does not travel with the repository)."""
    import shutil
    import subprocess
    exe = shutil.which("julia")
    if not exe:
        return "not available: no julia on PATH"
    script = ("using TFHE, Random; rng = MersenneTwister(123); sk, ck = make_key_pair(rng, tfhe_parameters_80()); "
              "x = encrypt(rng, sk, true); y = encrypt(rng, sk, false); gate_nand(ck, x, y); "
              "t = @elapsed for i in 1:5 gate_nand(ck, x, y) end; println(\"ms_per_gate=\", 1000 * t / 5)")
    try:
        r = subprocess.run([exe, "-e", script], capture_output=True, text=True, timeout=300)
    except (OSError, subprocess.TimeoutExpired) as e:
        return f"not available: {type(e).__name__}"
    if r.returncode != 0 or "ms_per_gate=" not in r.stdout:
        return "not available: julia is present but `using TFHE` failed"
    return {"ms_per_gate_single_thread": float(r.stdout.split("ms_per_gate=")[1].split()[0]), "what": "TFHE.jl gate_nand, tfhe_parameters_80, this box"}


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
        "cpu_model": cpu_model(),
        "julia_reference": julia_reference_probe(),
        "kind": "port",
        "sample": f"first {S} of the {hx.shape[0]} NAND gates of the GPU workload, one gate per OpenMP thread; "
                  f"C restatement of the reference algorithm (not Julia)",
        "single_thread_ms_per_gate": dt1 * 1e3,
        "ms_per_gate_under_load": threads / (S / dt) * 1e3,
        "note": (f"DRAM-bound at {threads} threads: every gate streams the 32.8 MB of key spectra and the 49 MB keyswitch key, so "
                 f"{threads} threads deliver {S / dt * dt1:.1f}x one thread, not {threads}x; value / this is not a compute ratio"),
        "parity_on_sample": bool(np.array_equal(want, gpu_out[:S])),
    }


if __name__ == "__main__":
    main()
