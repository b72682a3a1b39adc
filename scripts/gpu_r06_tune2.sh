#!/bin/bash
# round-6: clip-lanes verdicts of configs 3 / 4 (bench.py now tunes them for every clip workload), dumped for merge_tuned.py
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r06_tune2
mkdir -p $O
for w in cfg3 cfg4; do
  PTX_TUNE_ITERS=5 PTX_TUNED_OUT=$O/tuned_$w.json timeout 900 python bench.py --workload $w --steps 10 --warmup 3 --verbose --no-x3 --no-cpu-baseline > $O/bench_$w.log 2> $O/bench_$w.err; echo "bench $w exit $?"
  grep -h "^tune clip" $O/bench_$w.log
  python - <<PY
import json
for l in open("$O/bench_$w.log"):
    if l.startswith("{"):
        j = json.loads(l); cl = j["clip_lanes"]; print("$w:", j["value"], j["unit"], j["ms_per_step"], "lanes", j["config"].get("clip_lanes"), "| other leg:", cl and (cl["lanes"], cl["value"]))
PY
done
