#!/bin/bash
T=r04e; O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_circuit.py tests/test_bench_contract.py tests/test_gpu_parity.py tests/test_keygen.py -m gpu -q -x > $O/${T}_pytest.log 2>&1; rc=$?; tail -15 $O/${T}_pytest.log
if [ $rc -ne 0 ]; then echo "tests failed rc=$rc"; exit $rc; fi
timeout -k 10 300 python tools/circuit_timing.py > $O/${T}_circuit_timing.txt 2>&1; cat $O/${T}_circuit_timing.txt
