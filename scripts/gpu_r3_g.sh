#!/bin/bash
# Round-3 payload G: careful re-tune (PTX_TUNE_ITERS timed launches per candidate) of configs 2 / 3 (both precision legs),
# 1 / 4 / 5-fp32 and the zoo; dumps -> scripts/merge_tuned.py
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 PTX_TUNE_ITERS=${PTX_TUNE_ITERS:-6}
O=gpurun_out
PTX_TUNED_OUT=$O/r3g_tuned_cfg2.json timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r3g_bench_cfg2.json 2> $O/r3g_bench_cfg2.err
echo "cfg2 exit $? $(tail -1 $O/r3g_bench_cfg2.json | cut -c1-170)"
for w in cfg3 cfg4 cfg1 cfg5-fp32; do
  PTX_FULL_TUNE=1 PTX_TUNED_OUT=$O/r3g_tuned_$w.json timeout 900 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/r3g_bench_$w.json 2> $O/r3g_bench_$w.err
  echo "$w exit $? $(tail -1 $O/r3g_bench_$w.json | cut -c1-170)"
done
PTX_RETUNE=1 ZOO_NO_CPU=1 PTX_TUNED_OUT=$O/r3g_tuned_zoo.json timeout 1500 python scripts/gpu_zoo_bench.py > $O/r3g_zoo.log 2>&1; echo "zoo exit $?"; tail -18 $O/r3g_zoo.log
