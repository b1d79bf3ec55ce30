#!/usr/bin/env python3
"""Where a circuit level's time goes on the device: reads a rocprofv3 kernel trace (and memory-copy trace, if present) of
tools/circuit_timing.py and prints, for the steady-state levels, kernel durations and the idle gaps between consecutive
operations of the stream.
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/prof_circ -- python3 tools/circuit_timing.py
  python3 tools/level_gaps.py gpurun_out/prof_circ"""
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60]))
for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "memcpy " + r.get("Direction", "")))
rows.sort()
print(len(rows), "operations")
# a level = prologue ... up to the next prologue; take levels whose blind rotation is the single-rotation kernel
idx = [i for i, r in enumerate(rows) if r[2].startswith("prologue_kernel")]
levels = [rows[a:b] for a, b in zip(idx, idx[1:])]
steady = [lv for lv in levels if 1.0e6 < lv[-1][1] - lv[0][0] < 3.0e6][20:60]
if not steady:
    sys.exit("no steady-state levels found")
import statistics as st
print(f"{len(steady)} steady-state levels: median length {st.median(l[-1][1] - l[0][0] for l in steady) / 1e3:.1f} us (first op start to last op end), "
      f"median period {st.median(b[0][0] - a[0][0] for a, b in zip(steady, steady[1:])) / 1e3:.1f} us")
lv = steady[len(steady) // 2]
prev_end = None
for s, e, name in lv:
    gap = "" if prev_end is None else f"gap {(s - prev_end) / 1e3:7.1f} us  "
    print(f"  {gap:>18}{name:<62} {(e - s) / 1e3:8.1f} us")
    prev_end = e
nxt = levels[levels.index(lv) + 1][0]
print(f"  gap {(nxt[0] - prev_end) / 1e3:7.1f} us  (next level's first operation)")
busy = st.median(sum(e - s for s, e, _ in l) for l in steady) / 1e3
print(f"median busy time per level {busy:.1f} us")
