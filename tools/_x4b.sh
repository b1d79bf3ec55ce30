set -e; mkdir -p gpurun_out
for r in 1 2; do
for cfg in "1" "2host --gates 16" "2host --gates 256" "4a --gates 1" "4a --gates 128"; do
for v in base H3 H7; do
  unset TFHE_MI355X_LIB
  case $v in H*) export TFHE_MI355X_LIB=$PWD/tfhe.jl_amd/lib/libx_$v.so;; esac
  echo "== $cfg $v" >> gpurun_out/x4b.log
  timeout -k 10 120 python tools/run_config.py --config $cfg --reps 30 >> gpurun_out/x4b.log 2>> gpurun_out/x4b.err
done
done
done
python3 - <<'PY'
import json
lab=None
for l in open('gpurun_out/x4b.log'):
    if l.startswith('=='): lab=l.strip(); continue
    try: d=json.loads(l)
    except Exception: continue
    print(lab, round(d['blind_rotate_ms'],4), d.get('decrypt_ok_fraction'), d.get('kernel'))
PY
