#!/bin/bash
# round-4 session c: the parts of a split dispatch timed on the same device; the tail on a second stream; independent KATs on the GPU
T=r04c; O=gpurun_out
mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $O/valu_rates tools/probe/valu_rates.hip 2> $O/${T}_probe_build.log && timeout -k 10 120 $O/valu_rates > $O/${T}_valu_rates.txt 2>&1; cat $O/${T}_valu_rates.txt
timeout -k 10 900 python -m pytest tests/test_independent.py tests/test_gpu_parity.py -m gpu -q -x > $O/${T}_pytest.log 2>&1; rc=$?; tail -5 $O/${T}_pytest.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
timeout -k 10 600 python tools/sweep_sizes.py --params 80 --sizes 512,904,1024,2048,2560,3072,4096,5000,6400 --ab br_split=0 --ab br_split=2 --reps 7 > $O/${T}_split80.jsonl 2> $O/${T}_split80.err || { tail -5 $O/${T}_split80.err; exit 1; }
cat $O/${T}_split80.jsonl
timeout -k 10 600 python tools/sweep_sizes.py --params 128 --sizes 1024,2048,3072,5000 --ab br_split=0 --ab br_split=2 --reps 5 > $O/${T}_split128.jsonl 2> $O/${T}_split128.err || { tail -5 $O/${T}_split128.err; exit 1; }
cat $O/${T}_split128.jsonl
