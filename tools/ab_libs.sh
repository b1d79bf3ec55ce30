#!/bin/bash
# A/B of alternative builds of the library on ONE device: tools/ab_libs.sh <config> <rounds> lib1.so lib2.so ...
# (each run is a fresh process; `TFHE_MI355X_LIB` selects the build).  Prints blind-rotate ms and the decrypt check.
CFG=$1; ROUNDS=$2; shift 2
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    TFHE_MI355X_LIB=$PWD/$lib timeout -k 10 200 python tools/run_config.py --config $CFG --no-diag $EXTRA 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib $EXTRA', round(d['blind_rotate_ms'],3), d['decrypt_ok_fraction'])" || exit 1
  done
done
