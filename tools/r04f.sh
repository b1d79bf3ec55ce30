#!/bin/bash
T=r04f; O=gpurun_out
mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_circ -- python3 tools/circuit_timing.py > $O/${T}_trace.log 2>&1 || { tail -5 $O/${T}_trace.log; exit 1; }
python3 tools/level_gaps.py $O/prof_circ > $O/${T}_level_gaps.txt 2>&1; cat $O/${T}_level_gaps.txt
find $O/prof_circ -name "*.csv" | head; find $O/prof_circ -name "*kernel_trace.csv" -exec head -3 {} \;
