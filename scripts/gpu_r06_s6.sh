#!/bin/bash
# round-6 GPU session 6: analytic tap pruning (no workgroup OR-reduction in the generic conv prologue) -- kernel tests, A/B on
# configs 2 and 3 (PTX_PRUNE_ANALYTIC=1|0 alternating, --no-autotune: the shipped table on both arms, single plan)
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv and not f16 and not stem and not body and not tstack and not program" > gpurun_out/r06_s6_kernels.txt 2>&1; echo "kernel tests exit $?"; tail -n 2 gpurun_out/r06_s6_kernels.txt
for w in cfg2 cfg3; do
for v in 1 0 1 0; do
  PTX_PRUNE_ANALYTIC=$v PTX_BENCH_ROWS=gpurun_out/r06_rows_${w}_prune$v.txt timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-x3 --no-lanes --lanes 1 --no-cpu-baseline --no-autotune > gpurun_out/r06_bench_${w}_prune$v.log 2> gpurun_out/r06_bench_${w}_prune$v.err
  python - <<PY
import json
for l in open("gpurun_out/r06_bench_${w}_prune$v.log"):
    if l.startswith("{"):
        j = json.loads(l); print("$w analytic=$v:", j["value"], j["ms_per_step"], "plain pass", j["launch_timing"]["plain_pass_ms"])
PY
done
done
