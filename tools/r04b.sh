#!/bin/bash
# round-4 session b: VALU issue-cost probe; 4-party accumulators in LDS; split dispatch of part-filled rounds; k = 2 groups of three
T=r04b; O=gpurun_out
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/valu_rates tools/probe/valu_rates.hip 2> $O/${T}_probe_build.log && timeout -k 10 120 $O/valu_rates > $O/${T}_valu_rates.txt 2>&1; cat $O/${T}_valu_rates.txt
timeout -k 10 900 python -m pytest tests/test_kernel_matrix.py tests/test_mk.py -m gpu -q -x -k "mk or ragged or lockstep" > $O/${T}_pytest.log 2>&1; rc=$?; tail -5 $O/${T}_pytest.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
for rep in 1 2; do
  for acc in -1 1; do
    timeout -k 10 300 python tools/run_config.py --config mk4 --reps 4 --no-diag --set mkg_acc=$acc >> $O/${T}_mk4.jsonl 2>> $O/${T}_mk4.err || exit 1
  done
done
timeout -k 10 300 python tools/run_config.py --config mk4 --reps 4 --no-diag --set mkg_rw=2 >> $O/${T}_mk4.jsonl 2>> $O/${T}_mk4.err || exit 1
cat $O/${T}_mk4.jsonl
timeout -k 10 400 python tools/sweep_sizes.py --params 80 --sizes 1100,1536,2560,3072,4096,5000,6144,9900 --ab br_split=0 > $O/${T}_split80.jsonl 2> $O/${T}_split80.err || { tail -5 $O/${T}_split80.err; exit 1; }
cat $O/${T}_split80.jsonl
timeout -k 10 400 python tools/sweep_sizes.py --params 128 --sizes 3072,5000 --ab br_split=0 > $O/${T}_split128.jsonl 2> $O/${T}_split128.err || { tail -5 $O/${T}_split128.err; exit 1; }
cat $O/${T}_split128.jsonl
timeout -k 10 600 python tools/sweep_sizes.py --params k2 --sizes 1792,4096,7168,16384 --ab k2_rw=3 --ab k2_rw=1 --reps 3 > $O/${T}_k2.jsonl 2> $O/${T}_k2.err || { tail -5 $O/${T}_k2.err; exit 1; }
cat $O/${T}_k2.jsonl
