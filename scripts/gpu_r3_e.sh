#!/bin/bash
# Round-3 payload E: rocprofv3 kernel trace + PMC passes (tuned table loaded, no tuner launches) of config 2 and config 3 on
# the fp32 path, each followed by the bench line of the same build; summaries -> gpurun_out/prof_*/summary.txt
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
for w in cfg2 cfg3; do
  W=$w TAG=_r3 STEPS=10 bash scripts/gpu_prof_pmc.sh > $O/r3e_prof_$w.log 2>&1; tail -3 $O/r3e_prof_$w.log
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-x3 --no-autotune $( [ $w = cfg3 ] && echo --no-cpu-baseline ) > $O/r3e_bench_$w.json 2> $O/r3e_bench_$w.err
  echo "$w exit $? $(tail -1 $O/r3e_bench_$w.json | cut -c1-200)"
done
