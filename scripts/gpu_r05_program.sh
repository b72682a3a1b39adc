#!/bin/bash
# Round 5, conv programs: GPU parity tests, then A/B of the launch-per-conv plan vs the program plan on configs 2 and 3.
# Usage (GPU box): bash scripts/gpu_r05_program.sh [tag]
set -u
cd "$(dirname "$0")/.."
TAG=${1:-a}
OUT=gpurun_out/r05_program_$TAG
mkdir -p $OUT
export PTX_COMMIT=$(cat .commit_for_gpurun 2>/dev/null || echo unknown)
export PTX_PROG_SPIN_LIMIT=${PTX_PROG_SPIN_LIMIT:-1000000}
python -c "import torch; print(torch.cuda.get_device_name(0))" > $OUT/device.txt 2>&1
timeout 900 python -m pytest tests/test_conv_program.py -m gpu -x -q > $OUT/pytest_program.log 2>&1
RC=$?
echo "pytest rc=$RC" | tee -a $OUT/summary.txt
tail -25 $OUT/pytest_program.log
if [ $RC -ne 0 ]; then exit $RC; fi
run() {   # name workload env...
  local name=$1 wl=$2; shift 2
  env "$@" PTX_BENCH_ROWS=$OUT/rows_${name}.txt timeout 600 python bench.py --workload $wl --steps 30 --warmup 5 \
      --no-cpu-baseline --no-x3 --no-autotune > $OUT/bench_${name}.json 2> $OUT/bench_${name}.err
  python - "$OUT/bench_${name}.json" "$name" <<'PY' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print("%-28s %9.1f %s  %.4f ms/step  parity=%s" % (sys.argv[2], d["value"], d["unit"], d["ms_per_step"], d.get("parity")))
except Exception as e:
    print("%-28s FAILED %s" % (sys.argv[2], e))
PY
}
for wl in cfg2 cfg3; do
  run ${wl}_launches $wl PTX_PROGRAM=0
  run ${wl}_prog_w1 $wl PTX_PROGRAM_WGS=1
  run ${wl}_prog_w2 $wl PTX_PROGRAM_WGS=2
  run ${wl}_prog_w3 $wl PTX_PROGRAM_WGS=3
  run ${wl}_prog_w2_m16k $wl PTX_PROGRAM_WGS=2 PTX_PROGRAM_MAX_M=16384
  run ${wl}_prog_w2_tuned $wl PTX_PROGRAM_WGS=2 PTX_PROGRAM_TILES=tuned
done
grep -h 'chain conv_program\|conv_program' $OUT/rows_*prog_w2.txt | head -20 | tee -a $OUT/summary.txt
