#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into a short text summary (per-kernel time stats and per-kernel PMC
averages) and counters.json: per kernel the average duration, HBM bytes per launch (FETCH_SIZE x 1024 x 2 on gfx950 +
WRITE_SIZE x 1024, MI355X_MICROARCH.md §HBM), VALU instructions, VALU-busy fraction and the clock held
(GRBM_GUI_ACTIVE / 8 XCDs / duration) — what bench.py quotes under roofline.traffic / roofline_secondary."""
import csv, glob, json, os, sys, time
from collections import defaultdict

out = sys.argv[1]
cmd = sys.argv[2] if len(sys.argv) > 2 else ""
def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))

stats = {}
print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("stats/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        stats[row.get("Name", "")] = {"calls": int(row.get("Calls", 0) or 0), "avg_ns": float(row.get("AverageNs", 0) or 0)}
        print("  {:<70s} calls={:>5s} total_ns={:>14s} avg_ns={:>12s} pct={:>6s}".format(
            row.get("Name", "")[:70], row.get("Calls", ""), row.get("TotalDurationNs", ""),
            row.get("AverageNs", ""), row.get("Percentage", "")))
print("== kernel trace: per-dispatch resources ==")
grid = {}
seen = set()
durs = defaultdict(list)
for f in find("stats/**/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")
        try:
            durs[k].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
        except Exception:
            pass
        try:
            grid[k] = int(row.get("Grid_Size_X", row.get("Grid_Size", 0))) // max(1, int(row.get("Workgroup_Size_X", row.get("Workgroup_Size", 1))))
        except Exception:
            pass
        if k in seen: continue
        seen.add(k)
        print("  {:<70s} vgpr={} agpr={} sgpr={} lds={} scratch={} grid={} wg={}".format(
            k[:70], row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
            row.get("Scratch_Size"), row.get("Grid_Size_X", row.get("Grid_Size")), row.get("Workgroup_Size_X", row.get("Workgroup_Size"))))
print("== blind-rotate launches in time order (ms): the first launch of a process runs before clocks and caches have settled ==")
for k, v in durs.items():
    if "blind_rotate" in k and len(v) > 1:
        d = [x[1] / 1e6 for x in sorted(v)]
        print("  {:<70s} first={:.3f} others: mean={:.3f} min={:.3f} max={:.3f} (n={})".format(k[:70], d[0], sum(d[1:]) / len(d[1:]), min(d[1:]), max(d[1:]), len(d) - 1))
print("== PMC (average per dispatch) ==")
pmc = defaultdict(dict)
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        key = (row.get("Kernel_Name", ""), row.get("Counter_Name", ""))
        acc[key][0] += float(row.get("Counter_Value", 0) or 0)
        acc[key][1] += 1
    for (k, c), (s, n) in sorted(acc.items()):
        pmc[k][c] = s / n
        if "blind_rotate" in k or "keyswitch" in k:
            print("  {:<60s} {:<28s} avg={:.6g} (n={})".format(k[:60], c, s / n, n))

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from source_hash import kernel_source_sha16
counters = {"_meta": {"command": cmd, "date": time.strftime("%Y-%m-%d"), "kernel_source_sha16": kernel_source_sha16(),
                      "correction": "FETCH_SIZE x2 (gfx950 wide-load undercount), every --pmc group in its own run"}}
for k, c in pmc.items():
    if not ("blind_rotate" in k or "keyswitch" in k):
        continue
    e = {}
    if k in stats: e["avg_duration_ms"] = stats[k]["avg_ns"] / 1e6
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
        e["hbm_bytes_per_launch"] = c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024
    if "SQ_INSTS_VALU" in c: e["valu_insts_per_launch"] = c["SQ_INSTS_VALU"]
    if "GRBM_GUI_ACTIVE" in c:
        e["gui_active_cycles_per_xcd"] = c["GRBM_GUI_ACTIVE"] / 8
        if k in stats and stats[k]["avg_ns"] > 0:
            e["clock_mhz_from_gui_active"] = c["GRBM_GUI_ACTIVE"] / 8 / (stats[k]["avg_ns"] * 1e-9) / 1e6
        if "SQ_ACTIVE_INST_VALU" in c:
            # SQ_ACTIVE_INST_VALU counts quad-cycles summed over all waves; 1024 SIMDs x cycles of the launch available
            e["valu_busy_frac"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
        e["valu_active_per_wave_cycle"] = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        e["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    if k in grid: e["workgroups_per_launch"] = grid[k]      # blind-rotate kernels: one workgroup per rotation (per four: v3<...,4>)
    counters[k] = e
json.dump(counters, open(os.path.join(out, "counters.json"), "w"), indent=1)
print("== derived (counters.json) ==")
for k, e in counters.items():
    if k != "_meta": print("  ", k[:70], json.dumps(e))
