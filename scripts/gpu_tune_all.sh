#!/bin/bash
# Re-tune every BASELINE workload (both precision legs) and dump the tile tables: merge them into
# pretorched-x_amd/tuned_gfx950.json with scripts/merge_tuned.py.
export PYTHONDONTWRITEBYTECODE=1 PTX_FULL_TUNE=1
mkdir -p gpurun_out
for w in cfg2 cfg3 cfg4 cfg5 cfg1 cfg5-fp32; do
  PTX_TUNED_OUT=gpurun_out/tuned_$w.json timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/tune_$w.log 2>&1
  echo "$w exit $? $(tail -1 gpurun_out/tune_$w.log | cut -c1-120)"
done
