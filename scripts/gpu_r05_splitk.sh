#!/bin/bash
# Round 5: fused split-K (write-through partials + last-arriver reduce) against the reduce launch, configs 2 and 3; then a
# full re-tune with it on.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_splitk_${1:-a}
mkdir -p $OUT
export PTX_COMMIT=$(cat .commit_for_gpurun 2>/dev/null || echo unknown)
timeout 500 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "splitk or every_config or dual_source or geometries or chain or bit_exact" > $OUT/pytest_kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_kernels.log
timeout 300 python -m pytest tests/test_conv_program.py -m gpu -x -q > $OUT/pytest_program.log 2>&1; echo "program tests rc=$?" | tee -a $OUT/summary.txt; tail -3 $OUT/pytest_program.log
run() {
  local name=$1 wl=$2 extra=$3; shift 3
  env "$@" PTX_TUNE_LOG=$OUT/tune_${name}.log PTX_BENCH_ROWS=$OUT/rows_${name}.txt timeout 900 python bench.py --workload $wl --steps 30 --warmup 5 \
      --no-cpu-baseline --no-x3 $extra > $OUT/bench_${name}.json 2> $OUT/bench_${name}.err
  python - "$OUT/bench_${name}.json" "$name" <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    n = sum(v["launches"] for v in d["roofline_net"]["per_kernel"].values())
    print("%-24s %9.1f %s  %.4f ms/step  conv launches %d  frac %s issued %s" % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"], n,
          d["roofline_net"]["frac"], d["roofline_net"].get("issued_frac")))
except Exception as e:
    print("%-24s FAILED %s" % (sys.argv[2], e))
PY
}
for wl in cfg2 cfg3; do
  run ${wl}_reduce $wl --no-autotune PTX_SPLITK_FUSED=0 PTX_PROGRAM=0
  run ${wl}_fused $wl --no-autotune PTX_SPLITK_FUSED=1 PTX_PROGRAM=0
done
run cfg2_retune_fused cfg2 "" PTX_SPLITK_FUSED=1 PTX_PROGRAM=0 PTX_TUNED_OUT=$OUT/tuned_cfg2.json
run cfg3_retune_fused cfg3 "" PTX_SPLITK_FUSED=1 PTX_PROGRAM=0 PTX_FULL_TUNE=1 PTX_TUNED_OUT=$OUT/tuned_cfg3.json
run cfg2_retune_reduce cfg2 "" PTX_SPLITK_FUSED=0 PTX_PROGRAM=0 PTX_TUNED_OUT=$OUT/tuned_cfg2_reduce.json
