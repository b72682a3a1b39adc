#!/bin/bash
# round-6 GPU session 8: same-box A/B of the shipped tuned table against the same table with the 12 ks2 adoptions of session 7
# (bench.py --no-autotune --lanes 1, alternating, two passes per arm)
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r06_s8
mkdir -p $O
for w in cfg3 cfg2; do
for v in ks2 shipped ks2 shipped; do
  T=""; if [ $v = ks2 ]; then T="$(pwd)/scripts/tmp_tuned_ks2.json"; fi
  PTX_TUNED_TABLE=$T PTX_BENCH_ROWS=$O/rows_${w}_$v.txt timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-x3 --no-lanes --lanes 1 --no-cpu-baseline --no-autotune > $O/bench_${w}_$v.log 2> $O/bench_${w}_$v.err
  python - <<PY
import json
for l in open("$O/bench_${w}_$v.log"):
    if l.startswith("{"):
        j = json.loads(l); print("$w $v:", j["value"], j["ms_per_step"], "plain pass", j["launch_timing"]["plain_pass_ms"])
PY
done
done
