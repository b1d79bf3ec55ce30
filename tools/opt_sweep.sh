#!/bin/bash
# One configuration under several engine-option settings (each "name=value[,name=value...]" or "-" for defaults), one process each:
#   bash tools/opt_sweep.sh <tag> <config> <setting>...
TAG=$1; CFG=$2; shift 2
mkdir -p gpurun_out
for st in "$@" "$1"; do
  SETS=""; if [ "$st" != "-" ]; then for kv in ${st//,/ }; do SETS="$SETS --set $kv"; done; fi
  timeout -k 10 300 python tools/run_config.py --config $CFG --reps 7 --no-diag $SETS 2>> gpurun_out/${TAG}.err | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-8s %-36s %-44s BR %8.3f ms  decrypt_ok %.3f' % ('$CFG', '$st', d['kernel'], d['blind_rotate_ms'], d['decrypt_ok_fraction']))" | tee -a gpurun_out/${TAG}.txt || exit 1
done
