#!/bin/bash
# A/B of two builds of the library on ONE device, interleaved: bash tools/probe/ab_libs.sh <old.so> <new.so> [rounds] [bench args...]
# prints the blind-rotate launch time (HIP events inside bench.py) of every run
OLD=$1; NEW=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for L in $OLD $NEW; do
    TFHE_MI355X_LIB=$L timeout -k 10 300 python bench.py --no-cpu-baseline --no-diagnostics "$@" > /tmp/ab.json 2>/tmp/ab.err || { tail -3 /tmp/ab.err; exit 1; }
    python - "$L" <<'PY'
import json, sys
d = json.load(open("/tmp/ab.json"))
print(f"{sys.argv[1]:45s} launch {d['roofline']['avg_launch_ms']:.3f} ms  step {d['ms_per_step']:.3f} ms  {d['roofline']['kernel']}", flush=True)
PY
  done
done
