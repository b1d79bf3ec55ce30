#!/bin/bash
# A/B of two builds of the library on ONE device, alternating processes:
#   bash tools/ab_libs.sh <tag> <libA.so> <libB.so> [configs...]     (configs: run_config.py names, default "2host 4a 4b 5 k2 1")
# Prints the blind-rotate kernel time (HIP events, median of the runs) of every (config, lib, round).
TAG=$1; A=$2; B=$3; shift 3
CFGS=${@:-2host 4a 4b 5 k2 1}
OUT=gpurun_out/${TAG}_ab.jsonl
mkdir -p gpurun_out
for round in 1 2; do
  for c in $CFGS; do
    for lib in $A $B; do
      TFHE_MI355X_LIB=$lib timeout -k 10 300 python tools/run_config.py --config $c --reps 7 --no-diag 2>> gpurun_out/${TAG}_ab.err | \
        python -c "import sys, json; d = json.loads(sys.stdin.read()); d['lib'] = '$lib'; d['round'] = $round; print(json.dumps(d))" >> $OUT || exit 1
    done
  done
done
python - <<EOF
import json
rows = [json.loads(l) for l in open("$OUT")]
for r in rows:
    print(f"{r['config'][:44]:44s} {r['lib'].split('/')[-1]:28s} round {r['round']}  {r['kernel']:42s} BR {r['blind_rotate_ms']:8.3f} ms  KS {r['keyswitch_ms']:.3f} ms  frac {r.get('frac_hbm_algorithmic', 0):.3f}")
EOF
