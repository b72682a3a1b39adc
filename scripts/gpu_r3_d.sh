#!/bin/bash
# Round-3 payload D: re-time every conv problem of the whole zoo with this build's tiles (row-major-epilogue tiles, chained
# tiles, alt decisions) and dump the tables; merge with scripts/merge_tuned.py.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
echo "== zoo (fp32, retune)"
PTX_RETUNE=1 ZOO_NO_CPU=1 PTX_TUNED_OUT=$O/r3d_tuned_zoo.json timeout 1500 python scripts/gpu_zoo_bench.py > $O/r3d_zoo.log 2>&1; echo "exit $?"
cp $O/zoo_bench.json $O/r3d_zoo_bench_retune.json 2>/dev/null
tail -20 $O/r3d_zoo.log
for w in cfg1 cfg4; do
  PTX_FULL_TUNE=1 PTX_TUNED_OUT=$O/r3d_tuned_$w.json timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-x3 --no-cpu-baseline > $O/r3d_bench_$w.json 2> $O/r3d_bench_$w.err
  echo "$w exit $? $(tail -1 $O/r3d_bench_$w.json | cut -c1-160)"
done
