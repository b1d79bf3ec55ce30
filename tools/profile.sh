#!/bin/bash
# Usage (on the GPU box, from the repo root):
#   bash tools/profile.sh <tag>                                   profiles `python3 bench.py --gpus 1 --steps 20 --warmup 5` (the
#                                                                 driver's command), without the CPU baseline and the DIAG runs
#   bash tools/profile.sh <tag> tools/run_config.py --config 4a   profiles any other python program
# Writes rocprofv3 kernel-trace stats and PMC passes (each in its own run, program directly after `--`) under
# gpurun_out/prof_<tag>/ and a summary (summary.txt, counters.json, kernel_stats.csv) to copy into profiles/.
set -o pipefail
TAG=${1:-run}; shift
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ $# -eq 0 ]; then
  CMD="bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-diagnostics"
else
  CMD="$@ --no-diag"
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $CMD > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
for PMC in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL"; do
  NAME=$(echo $PMC | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc_$NAME -- python3 $CMD > $OUT/pmc_$NAME.log 2>&1 || { echo "pmc pass $PMC failed"; tail -5 $OUT/pmc_$NAME.log; }
done
python3 tools/prof_summary.py $OUT "$CMD" > $OUT/summary.txt 2>&1
cp $(find $OUT/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv 2>/dev/null
cat $OUT/summary.txt
