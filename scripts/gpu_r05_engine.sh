#!/bin/bash
# Round 5: conv programs inside the engine (tuner A/B, forced on / off), configs 2 and 3; optional x3 PMC profile.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_engine_${1:-a}
mkdir -p $OUT
export PTX_COMMIT=$(cat .commit_for_gpurun 2>/dev/null || echo unknown)
timeout 300 python -m pytest tests/test_conv_program.py tests/test_boundary.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/summary.txt; tail -4 $OUT/pytest.log
run() {
  local name=$1 wl=$2; shift 2
  env "$@" PTX_TUNE_LOG=$OUT/tune_${name}.log PTX_BENCH_ROWS=$OUT/rows_${name}.txt timeout 600 python bench.py --workload $wl --steps 30 --warmup 5 \
      --no-cpu-baseline --no-x3 --no-autotune > $OUT/bench_${name}.json 2> $OUT/bench_${name}.err
  python - "$OUT/bench_${name}.json" "$name" <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    n = sum(v["launches"] for v in d["roofline_net"]["per_kernel"].values())
    print("%-24s %9.1f %s  %.4f ms/step  conv launches %d  issued_frac %s  frac %s" % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"], n,
          d["roofline_net"].get("issued_frac"), d["roofline_net"]["frac"]))
except Exception as e:
    print("%-24s FAILED %s" % (sys.argv[2], e))
PY
}
for wl in ${WLS:-cfg2 cfg3}; do
  run ${wl}_off $wl PTX_PROGRAM=0
  run ${wl}_auto $wl PTX_PROGRAM=auto
  run ${wl}_force $wl PTX_PROGRAM=force
  run ${wl}_force_m16k $wl PTX_PROGRAM=force PTX_PROGRAM_MAX_M=16384
done
grep -h 'program' $OUT/tune_*auto.log | head -12 | tee -a $OUT/summary.txt
grep -h 'conv_program' $OUT/rows_*force.txt | tee -a $OUT/summary.txt
