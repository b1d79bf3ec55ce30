#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into a short text summary (per-kernel time stats and
per-kernel PMC averages)."""
import csv, glob, os, sys
from collections import defaultdict

out = sys.argv[1]
def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))

print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in find("stats/**/*kernel_stats.csv"):
    for row in csv.DictReader(open(f)):
        print("  {:<60s} calls={:>5s} total_ns={:>14s} avg_ns={:>12s} pct={:>6s}".format(
            row.get("Name", "")[:60], row.get("Calls", ""), row.get("TotalDurationNs", ""),
            row.get("AverageNs", ""), row.get("Percentage", "")))
print("== kernel trace: per-dispatch resources ==")
seen = set()
for f in find("stats/**/*kernel_trace.csv"):
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "")[:60]
        if k in seen: continue
        seen.add(k)
        print("  {:<60s} vgpr={} agpr={} sgpr={} lds={} scratch={} grid={} wg={}".format(
            k, row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("SGPR_Count"), row.get("LDS_Block_Size"),
            row.get("Scratch_Size"), row.get("Grid_Size_X", row.get("Grid_Size")), row.get("Workgroup_Size_X", row.get("Workgroup_Size"))))
print("== PMC (average per dispatch) ==")
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        key = (row.get("Kernel_Name", "")[:50], row.get("Counter_Name", ""))
        acc[key][0] += float(row.get("Counter_Value", 0) or 0)
        acc[key][1] += 1
    for (k, c), (s, n) in sorted(acc.items()):
        if "blind_rotate" in k or "keyswitch" in k:
            print("  {:<50s} {:<28s} avg={:.6g} (n={})".format(k, c, s / n, n))

# HBM traffic per launch for every kernel with both counters (MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are
# in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads -> x2).
import json
fs, ws = {}, {}
for f in find("pmc_*/**/*counter_collection.csv"):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        if row.get("Counter_Name") in ("FETCH_SIZE", "WRITE_SIZE"):
            key = (row.get("Kernel_Name", ""), row["Counter_Name"])
            acc[key][0] += float(row.get("Counter_Value", 0) or 0); acc[key][1] += 1
    for (k, c), (sm, n) in acc.items():
        (fs if c == "FETCH_SIZE" else ws)[k] = sm / n
traffic = {}
for k in fs:
    if k in ws:
        traffic[k] = {"FETCH_SIZE_KiB": fs[k], "WRITE_SIZE_KiB": ws[k],
                      "hbm_bytes_per_launch": fs[k] * 1024 * 2 + ws[k] * 1024,
                      "correction": "FETCH_SIZE x2 (gfx950 wide-load undercount), separate --pmc passes"}
json.dump(traffic, open(os.path.join(out, "traffic.json"), "w"), indent=1)
