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
    # the kernel named is the one the engine launched for this batch size (512 rotations: the two-wave kernel)
    assert rf["kernel"] == "blind_rotate_kernel_w2<2,rw2>" and "bound_note" in rf and "traffic_replayed_from" in rf and "traffic_is_replayed" in rf
    r2 = d["roofline_secondary"]
    assert r2["bound"] == "fp64_valu" and r2["kernel"] == rf["kernel"] and r2["flops_per_unit"] == 500 * (6 * 5 * 512 * 9 + 8 * 512 * 8)
    assert abs(r2["achieved"] - 512 * r2["flops_per_unit"] / (rf["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * r2["achieved"]
    assert 300 < r2["clock_mhz"] < 2600 and 0 < r2["rounding_margin"] < 0.25
    assert abs(r2["frac_at_measured_clock"] - r2["frac"] * 2400 / r2["clock_mhz"]) < 1e-9
    assert 0 < d["value_pcie_inclusive"] < 1.2 * d["value"]


@pytest.mark.gpu
def test_bench_two_rank_launch_rehearsal():
    """The driver's multi-GPU launch line with 2 ranks, rehearsed on one GPU (TFHE_BENCH_SHARE_GPU=1: both ranks on
    device 0, gloo instead of RCCL for barrier / max-reduce / gather): rank 0 prints one whole-job JSON line."""
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


def _one_json_line(cmd, env=None):
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_spawns_its_own_ranks_and_gathers():
    """`python bench.py --gpus 2` with no launcher: the parent starts the two rank processes itself (before touching a
    GPU) and relays rank 0's line; the shards' results are gathered to rank 0 (one-GPU rehearsal: gloo on host copies)
    and rank 0's own shard is found in the gathered tensor.  Mixed workload: uneven, rotation-balanced shards."""
    env = dict(os.environ, TFHE_BENCH_SHARE_GPU="1")
    env.pop("WORLD_SIZE", None)
    d = _one_json_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gates", "300"], env)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["outputs_decrypt_correctly"] is True
    assert d["gather_matches_local_shard"] is True and "gather to rank 0" in d["config"]["result_gather"]
    assert abs(d["value"] - 2 * 300 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    d = _one_json_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "mixed"], env)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["outputs_decrypt_correctly"] is True
    assert d["gather_matches_local_shard"] is True
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_fanout_context():
    """`--fanout`: one process, one multi-device context (two device contexts on this one GPU): the library splits the
    host batch itself; the line says that inputs crossed PCIe inside the timed region."""
    env = dict(os.environ, TFHE_BENCH_SHARE_GPU="1")
    d = _one_json_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--fanout", "--steps", "2", "--warmup", "1", "--gates", "300"], env)
    assert d["n_gpus"] == 2 and d["outputs_decrypt_correctly"] is True and "multi-device context" in d["config"]["launch"]
    assert d["roofline"]["units_per_launch"] == 600 and "host buffers" in d["config"]["inputs"]
    assert abs(d["value"] - 600 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
def test_bench_single_rank_through_rccl():
    """TFHE_BENCH_FORCE_DIST=1: one rank, but the barrier, the max-reduce of the elapsed time and the result gather go
    through the RCCL process group (the only way to execute bench.py's RCCL calls on a one-GPU box)."""
    env = dict(os.environ, TFHE_BENCH_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TFHE_BENCH_SHARE_GPU"):
        env.pop(k, None)
    d = _one_json_line([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--gates", "300",
                        "--no-cpu-baseline", "--no-diagnostics"], env)
    assert d["n_gpus"] == 1 and d["outputs_decrypt_correctly"] is True and d["gather_matches_local_shard"] is True
    assert d["config"]["result_gather"].startswith("rccl gather to rank 0")


def test_committed_profile_matches_the_sources_and_the_instruction_budget():
    """CPU-side guard (no GPU): the rocprofv3 profile bench.py replays its PMC fields from must belong to the committed kernel
    sources (tools/source_hash.py), and the headline kernel must not have grown — 4.273 G vector instructions per 4096 x 500
    rotation-steps since round 3 (2086 per wave-step).  A change to the l = 2 instantiation that only moved the launch time
    by 1 % once slipped through three interleaved A/Bs as noise; its instruction count did not."""
    import glob
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_hash import kernel_source_sha16
    want = kernel_source_sha16(ROOT)
    hits = []
    for f in glob.glob(os.path.join(ROOT, "profiles", "*_cfg2", "counters.json")):
        d = json.load(open(f))
        if d.get("_meta", {}).get("kernel_source_sha16") == want:
            hits.append((f, d))
    assert hits, f"no profiles/*_cfg2/counters.json for kernel sources {want}: run `bash tools/gpu_session.sh <tag> bench prof2` and commit it"
    for f, d in hits:
        v3 = [v for k, v in d.items() if "blind_rotate_kernel_v3<2, 8, true, false, 4>" in k]
        assert v3, f
        assert v3[0]["valu_insts_per_launch"] <= 4.28e9, (f, v3[0]["valu_insts_per_launch"])


# ---- two DIFFERENT devices: collected everywhere, skipped until a box shows two (tests/conftest.py: two_gpus) ---------------------
from conftest import two_gpus  # noqa: E402


def _clean_env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "TFHE_BENCH_SHARE_GPU", "TFHE_BENCH_FORCE_DIST"):
        env.pop(k, None)
    return env


def _check_two_gpu_line(d, gates=None):
    assert d["n_gpus"] == 2 and d["outputs_decrypt_correctly"] is True
    assert d["gather_matches_local_shard"] is True and d["gather_matches_every_ranks_checksum"] is True and d["every_rank_decrypts"] is True
    g = d["config"]["result_gather"]
    # real RCCL between two GPUs: the point-to-point gather to rank 0, or — if this RCCL build rejects `gather` — the all_gather
    # fallback, named as such (bench.py: ResultGather)
    assert g.startswith("rccl ") and ("gather to rank 0" in g) and "rehearsal" not in g, g
    if gates:
        assert abs(d["value"] - 2 * gates * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
@two_gpus
def test_bench_two_gpus_self_spawned_real_rccl():
    """`python bench.py --gpus 2` on a box with two GPUs: one rank per device, RCCL for the barrier, the max-reduce and the result
    gather — the first execution of bench.py's collective path between two different devices."""
    d = _one_json_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gates", "300"], _clean_env())
    assert d["scaling"] == "weak"
    _check_two_gpu_line(d, 300)
    d = _one_json_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "mixed"], _clean_env())
    assert d["scaling"] == "strong"
    _check_two_gpu_line(d)
    assert abs(d["value"] - 65536 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.gpu
@two_gpus
def test_bench_two_gpus_under_the_drivers_launcher():
    """The driver's own launch line (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`) on two GPUs."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = _one_json_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--gates", "512"], _clean_env())
    assert d["scaling"] == "weak" and "cpu_baseline" not in d
    _check_two_gpu_line(d, 512)


@pytest.mark.gpu
@two_gpus
def test_bench_fanout_on_two_gpus():
    """`--fanout --gpus 2`: ONE process, one multi-device context on devices {0, 1} (tfhe_ctx_create_multi with distinct ids)."""
    d = _one_json_line([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--fanout", "--steps", "2", "--warmup", "1", "--gates", "300"], _clean_env())
    assert d["n_gpus"] == 2 and d["outputs_decrypt_correctly"] is True and "multi-device context" in d["config"]["launch"]
    assert d["fanout_matches_one_device"] is True
    assert d["roofline"]["units_per_launch"] == 600 and "host buffers" in d["config"]["inputs"]
