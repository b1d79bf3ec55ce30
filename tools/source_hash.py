#!/usr/bin/env python3
"""sha256 (first 16 hex digits) of the sources libtfhe_mi355x.so is built from: stamps a rocprofv3 profile
(tools/prof_summary.py) and is recomputed by bench.py, which quotes PMC-derived counters only from a profile of THIS code."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha16(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "tfhe.jl_amd", "csrc", "*.hpp")) + glob.glob(os.path.join(root, "tfhe.jl_amd", "csrc", "*.hip"))
                   + glob.glob(os.path.join(root, "tfhe.jl_amd", "csrc", "Makefile")) + glob.glob(os.path.join(root, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_sha16())
