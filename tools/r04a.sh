#!/bin/bash
# round-4 session a: the N = 2048 kernel with exchanged rotated words against the round-3 kernel
T=r04a; O=gpurun_out
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernel_matrix.py -m gpu -q -x -k "n2048" > $O/${T}_pytest.log 2>&1; rc=$?; tail -5 $O/${T}_pytest.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
for rep in 1 2; do
  for v in 1 2; do
    timeout -k 10 300 python tools/run_config.py --config 4b --reps 4 --no-diag --set n2048_variant=$v >> $O/${T}_4b.jsonl 2>> $O/${T}_4b.err || exit 1
  done
done
for rw in 1 4; do
  timeout -k 10 300 python tools/run_config.py --config 4b --reps 4 --no-diag --set n2048_variant=2 --set n2048_rw=$rw >> $O/${T}_4b.jsonl 2>> $O/${T}_4b.err || exit 1
done
cat $O/${T}_4b.jsonl
for v in 1 2; do
  TFHE_MI355X_LIB=tfhe.jl_amd/lib/libtfhe_mi355x_stamp.so timeout -k 10 300 python tools/phase_profile.py --config 4b --set n2048_variant=$v >> $O/${T}_phase.txt 2>&1 || exit 1
done
cat $O/${T}_phase.txt
