#!/bin/bash
# One GPU-box session: tests, bench line, other configs, profiles.  Usage: bash tools/gpu_session.sh <tag> [steps...]
# steps (default all): test bench configs prof2 prof4a prof4b prof5 (also: profmk4 profmk8 testall sweep circuit)
# Every step runs under its own `timeout -k 10`; a step that times out or is killed ends the session (no further GPU step).
TAG=${1:-run}; shift
STEPS=${@:-test bench configs prof2 prof4a prof4b prof5}
mkdir -p gpurun_out
run() {  # run <seconds> <name> <command...>
  local secs=$1 name=$2; shift 2
  echo "== $name: $(date +%T)"
  timeout -k 10 $secs "$@"
  local rc=$?
  echo "== $name rc=$rc $(date +%T)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out / was killed: stopping the session"; exit $rc; fi
  return $rc
}
for s in $STEPS; do
  case $s in
    test)    run 900 pytest bash -c "python -m pytest tests -m gpu -q -x --durations=15 > gpurun_out/${TAG}_pytest.log 2>&1"; tail -30 gpurun_out/${TAG}_pytest.log ;;
    testall) run 1000 pytest bash -c "python -m pytest tests -m gpu -q --durations=15 > gpurun_out/${TAG}_pytest.log 2>&1"; tail -40 gpurun_out/${TAG}_pytest.log ;;
    bench)   run 400 bench bash -c "python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err"; cat gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err ;;
    configs) for c in 1 2host 3 4a 4b 5 k2 mk4 mk8; do run 300 config_$c bash -c "python tools/run_config.py --config $c >> gpurun_out/${TAG}_configs.jsonl 2>> gpurun_out/${TAG}_configs.err"; done; cat gpurun_out/${TAG}_configs.jsonl ;;
    prof2)   run 600 prof2 bash tools/profile.sh ${TAG}_cfg2 > gpurun_out/${TAG}_prof2.log 2>&1; tail -25 gpurun_out/${TAG}_prof2.log ;;
    prof4a)  run 600 prof4a bash tools/profile.sh ${TAG}_cfg4a tools/run_config.py --config 4a --reps 9 > gpurun_out/${TAG}_prof4a.log 2>&1; tail -12 gpurun_out/${TAG}_prof4a.log ;;
    prof4b)  run 600 prof4b bash tools/profile.sh ${TAG}_cfg4b tools/run_config.py --config 4b --reps 9 > gpurun_out/${TAG}_prof4b.log 2>&1; tail -12 gpurun_out/${TAG}_prof4b.log ;;
    prof5)   run 600 prof5 bash tools/profile.sh ${TAG}_cfg5 tools/run_config.py --config 5 --reps 9 > gpurun_out/${TAG}_prof5.log 2>&1; tail -12 gpurun_out/${TAG}_prof5.log ;;
    profmk4) run 600 profmk4 bash tools/profile.sh ${TAG}_mk4 tools/run_config.py --config mk4 --reps 9 > gpurun_out/${TAG}_profmk4.log 2>&1; tail -12 gpurun_out/${TAG}_profmk4.log ;;
    profmk8) run 900 profmk8 bash tools/profile.sh ${TAG}_mk8 tools/run_config.py --config mk8 --reps 6 > gpurun_out/${TAG}_profmk8.log 2>&1; tail -12 gpurun_out/${TAG}_profmk8.log ;;
    sweep)   run 400 sweep80 bash -c "python tools/dispatch_sweep.py --params 80 > gpurun_out/${TAG}_dispatch_sweep_80bit.md 2>&1"; run 400 sweep128 bash -c "python tools/dispatch_sweep.py --params 128 > gpurun_out/${TAG}_dispatch_sweep_128bit.md 2>&1"; cat gpurun_out/${TAG}_dispatch_sweep_80bit.md gpurun_out/${TAG}_dispatch_sweep_128bit.md ;;
    circuit) run 300 circuit bash -c "python tools/circuit_timing.py > gpurun_out/${TAG}_circuit_timing.txt 2>&1"; cat gpurun_out/${TAG}_circuit_timing.txt ;;
    *) run 600 custom bash -c "$s" ;;
  esac
done
echo "session done"
