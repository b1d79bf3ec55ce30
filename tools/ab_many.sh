#!/bin/bash
# Times one run_config.py configuration under several builds of the library, one process each, the first one repeated at the
# end (drift check):   bash tools/ab_many.sh <tag> <config> <lib>...
TAG=$1; CFG=$2; shift 2
mkdir -p gpurun_out
for lib in "$@" "$1"; do
  TFHE_MI355X_LIB=$lib timeout -k 10 300 python tools/run_config.py --config $CFG --reps 7 --no-diag 2>> gpurun_out/${TAG}.err | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-36s %-40s BR %8.3f ms  decrypt_ok %.3f' % ('$lib'.split('/')[-1], d['kernel'], d['blind_rotate_ms'], d['decrypt_ok_fraction']))" | tee -a gpurun_out/${TAG}.txt || exit 1
done
