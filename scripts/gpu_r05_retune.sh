#!/bin/bash
# careful re-tune of configs 2 and 3 on ONE box: shipped table vs a table tuned here with more iterations per candidate
cd "$(dirname "$0")/.."
O=gpurun_out/r05_retune
mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-26s %9.1f %s  %.4f ms/step" % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"]))
except Exception as e:
    print("%-26s FAILED %s" % (sys.argv[2], e))
PY
}
for wl in cfg2 cfg3; do
  for i in 1 2; do
    timeout 600 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --no-x3 --no-autotune > $O/${wl}_shipped_$i.json 2>/dev/null; line $O/${wl}_shipped_$i.json ${wl}_shipped_$i | tee -a $O/summary.txt
  done
  PTX_TUNE_ITERS=${ITERS:-6} PTX_FULL_TUNE=1 PTX_TUNED_OUT=$O/tuned_$wl.json timeout 900 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --no-x3 > $O/${wl}_tuning.json 2>/dev/null; line $O/${wl}_tuning.json ${wl}_tuning_run | tee -a $O/summary.txt
  for i in 1 2; do
    PTX_TUNED_TABLE=$O/tuned_$wl.json timeout 600 python bench.py --workload $wl --steps 40 --warmup 5 --no-cpu-baseline --no-x3 --no-autotune > $O/${wl}_retuned_$i.json 2>/dev/null; line $O/${wl}_retuned_$i.json ${wl}_retuned_$i | tee -a $O/summary.txt
  done
done
