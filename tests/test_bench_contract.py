"""bench.py contract: one JSON line with the required fields and self-consistent roofline numbers
(a regression here — e.g. the rotation count read after a later 1-gate call — silently corrupts `roofline.frac`)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line_is_consistent():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--gates", "512",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in d
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["outputs_decrypt_correctly"] is True
    rf = d["roofline"]
    assert rf["units_per_launch"] == 512 and rf["bytes_per_unit"] == 16384000
    assert abs(rf["achieved"] - 512 * 16384000 / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.01 < rf["frac"] < 1.5
    assert abs(d["value"] - 512 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    assert 0.1 < d["ms_per_bootstrap_single_gate"] < 50


@pytest.mark.gpu
def test_bench_two_rank_launch_rehearsal():
    """The driver's multi-GPU launch line with 2 ranks, rehearsed on one GPU (TFHE_BENCH_SHARE_GPU=1: both ranks on
    device 0, gloo instead of RCCL for barrier / max-reduce, no gather): rank 0 prints one whole-job JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, TFHE_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--warmup", "1", "--gates", "512"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "cpu_baseline" not in d
    assert abs(d["value"] - 2 * 512 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]   # whole-job aggregate
    assert d["outputs_decrypt_correctly"] is True
