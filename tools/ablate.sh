#!/bin/bash
# Timing-only experiment: libraries built with -DTFHE_ABL_* (parts of blind_rotate_kernel_v3 removed; results are wrong)
# against the real one, config 2 (4096 NAND, 80-bit set).   bash tools/ablate.sh <tag>
TAG=$1
mkdir -p gpurun_out
for lib in tfhe.jl_amd/lib/libtfhe_mi355x.so tfhe.jl_amd/lib/libabl_*.so tfhe.jl_amd/lib/libtfhe_mi355x.so; do
  TFHE_MI355X_LIB=$lib timeout -k 10 200 python tools/run_config.py --config 2host --reps 7 --no-diag 2>> gpurun_out/${TAG}_abl.err | \
    python -c "import sys, json; d = json.loads(sys.stdin.read()); print('%-44s BR %8.3f ms  decrypt_ok %.3f' % ('$lib'.split('/')[-1], d['blind_rotate_ms'], d['decrypt_ok_fraction']))" | tee -a gpurun_out/${TAG}_abl.txt || exit 1
done
