#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (tfhe.jl_amd/build/resource_usage.txt)."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else "tfhe.jl_amd/build/resource_usage.txt"
txt = open(path).read()
keys = [("VGPR", r"VGPRs"), ("AGPR", r"AGPRs"), ("SGPR", r"TotalSGPRs"), ("spill", r"VGPR Spill"),
        ("scratch", r"ScratchSize \[bytes/lane\]"), ("occ", r"Occupancy \[waves/SIMD\]"),
        ("LDS", r"LDS Size \[bytes/block\]")]
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    vals = []
    for label, k in keys:
        m = re.search(k + r": (\S+)", b)
        vals.append(f"{label}={m.group(1) if m else '?'}")
    print(f"{name[:58]:58s} " + " ".join(vals))
