#!/bin/bash
# the 32x32x2-MFMA strip tiles for the (2+1)D mid widths: correctness on every config, then a targeted tuning session on
# config 3 (candidates timed next to the table's incumbents; 2 % bar), then the same box with and without the new entries
cd "$(dirname "$0")/.."
O=gpurun_out/r05_strip
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "every_config or bit_exact or geometries" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt; tail -2 $O/pytest.log
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-26s %9.1f %s  %.4f ms/step" % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"]))
except Exception as e:
    print("%-26s FAILED %s" % (sys.argv[2], e))
PY
}
timeout 600 python bench.py --workload cfg3 --steps 40 --warmup 5 --no-cpu-baseline --no-x3 --no-autotune > $O/cfg3_shipped.json 2>/dev/null; line $O/cfg3_shipped.json cfg3_shipped | tee -a $O/summary.txt
PTX_TUNE_CANDIDATES="128x96x16/4x1,128x160x16/4x1,128x192x16/4x1" PTX_TUNE_ITERS=6 PTX_FULL_TUNE=1 PTX_TUNE_LOG=$O/tune.log PTX_TUNED_OUT=$O/tuned_cfg3.json \
  timeout 900 python bench.py --workload cfg3 --steps 40 --warmup 5 --no-cpu-baseline --no-x3 --verbose > $O/cfg3_tuning.log 2>/dev/null; line $O/cfg3_tuning.log cfg3_tuning_run | tee -a $O/summary.txt
PTX_TUNED_TABLE=$O/tuned_cfg3.json timeout 600 python bench.py --workload cfg3 --steps 40 --warmup 5 --no-cpu-baseline --no-x3 --no-autotune > $O/cfg3_new.json 2>/dev/null; line $O/cfg3_new.json cfg3_new_table | tee -a $O/summary.txt
grep -h '128x96x16/4x1\|128x160x16/4x1\|128x192x16/4x1' $O/tune.log | sort -t$'\t' -k1,1 | head -60 | tee -a $O/summary.txt
grep '^tune' $O/cfg3_tuning.log | grep '4x1/m32' | head -20 | tee -a $O/summary.txt
