#!/bin/bash
# Blind-rotate kernel time of several br_variant values over batch sizes, one library:  bash tools/sweep_variants.sh <tag> <lib> "<variants>" "<sizes>"
TAG=$1; LIB=$2; VARS=$3; SIZES=$4
mkdir -p gpurun_out
for b in $SIZES; do for v in $VARS; do
  TFHE_MI355X_LIB=$LIB timeout -k 10 300 python tools/run_config.py --config 2host --gates $b --reps 7 --no-diag --set br_small=-1 --set br_tiny=-1 --set br_variant=$v 2>> gpurun_out/${TAG}.err | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%6d gates  variant %s  %-40s BR %8.3f ms  decrypt_ok %.3f' % ($b, '$v', d['kernel'], d['blind_rotate_ms'], d['decrypt_ok_fraction']))" | tee -a gpurun_out/${TAG}.txt || exit 1
done; done
