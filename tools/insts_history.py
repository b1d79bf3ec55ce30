#!/usr/bin/env python3
"""Vector instructions per launch (SQ_INSTS_VALU) and average launch time of every blind-rotate kernel in every committed profile
(profiles/*/counters.json), in session order: a kernel's instruction count only moves when its code does, so a change that hides
inside the run-to-run spread of an A/B shows up here (the +1.05 % of sessions r09e - r09i did).
  python tools/insts_history.py > profiles/insts_per_launch_history.txt"""
import glob, json, os
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
rows = {}
for f in sorted(glob.glob(os.path.join(root, "*", "counters.json"))):
    d = json.load(open(f))
    tag = os.path.basename(os.path.dirname(f))
    for k, v in d.items():
        if "blind_rotate" in k and isinstance(v, dict) and v.get("valu_insts_per_launch", 0) > 1e8:
            rows.setdefault(k.split("(")[0], []).append((tag, v["valu_insts_per_launch"] / 1e9, v["avg_duration_ms"]))
for k, v in rows.items():
    print(k)
    for tag, g, ms in v:
        print(f"    {tag:22s} {g:9.4f} G vector instructions per launch   {ms:8.2f} ms")
