#!/bin/bash
# round-6 GPU session 7: the K-split (ks2) tiles -- full re-tune of configs 3 and 2 (6 iterations, the incumbent defends itself),
# tables dumped for merge_tuned.py; before / after lines with --no-autotune come from the next session
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r06_s7
mkdir -p $O
for w in cfg3 cfg2; do
  PTX_FULL_TUNE=1 PTX_TUNE_ITERS=6 PTX_TUNED_OUT=$O/tuned_$w.json timeout 1500 python bench.py --workload $w --steps 20 --warmup 5 --no-x3 --no-cpu-baseline --no-lanes --lanes 1 > $O/bench_$w.log 2> $O/bench_$w.err; echo "bench $w exit $?"
  python - <<PY
import json
for l in open("$O/bench_$w.log"):
    if l.startswith("{"):
        j = json.loads(l); print("$w (re-tuned in run):", j["value"], j["ms_per_step"])
PY
done
