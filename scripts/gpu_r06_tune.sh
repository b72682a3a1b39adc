#!/bin/bash
# round-6: tune what the shipped table does not hold yet (body / T-stacked verdicts, clip-lane verdicts, new problems) for every
# BASELINE configuration and dump the tables (PTX_TUNED_OUT) for scripts/merge_tuned.py
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r06_tune
mkdir -p $O
for w in cfg2 cfg3 cfg4 cfg5 cfg1; do
  PTX_TUNE_ITERS=5 PTX_TUNED_OUT=$O/tuned_$w.json timeout 900 python bench.py --workload $w --steps 10 --warmup 3 --verbose --no-x3 --no-cpu-baseline > $O/bench_$w.log 2> $O/bench_$w.err; echo "bench $w exit $?"
  grep -h "^tune .*body\|lanes" $O/bench_$w.log | head -24
  python - <<PY
import json
for l in open("$O/bench_$w.log"):
    if l.startswith("{"):
        j = json.loads(l); print("$w:", j["value"], j["unit"], j["ms_per_step"], "lanes", j["config"].get("clip_lanes"))
PY
done
