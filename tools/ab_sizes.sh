#!/bin/bash
# Blind-rotate kernel time of several library builds over batch sizes of one configuration:  bash tools/ab_sizes.sh <tag> <config> "<sizes>" <lib>...
TAG=$1; CFG=$2; SIZES=$3; shift 3
mkdir -p gpurun_out
for b in $SIZES; do for lib in "$@"; do
  TFHE_MI355X_LIB=$lib timeout -k 10 300 python tools/run_config.py --config $CFG --gates $b --reps 7 --no-diag --set br_small=-1 --set br_tiny=-1 2>> gpurun_out/${TAG}.err | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-10s %6d gates  %-28s %-40s BR %8.3f ms  decrypt_ok %.3f' % ('$CFG', $b, '$lib'.split('/')[-1], d['kernel'], d['blind_rotate_ms'], d['decrypt_ok_fraction']))" | tee -a gpurun_out/${TAG}.txt || exit 1
done; done
