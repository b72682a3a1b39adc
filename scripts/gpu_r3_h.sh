#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w scripts/micro/hbm_mix.hip -o /tmp/hbm_mix && /tmp/hbm_mix > $O/r3h_hbm_mix.log 2>&1; cat $O/r3h_hbm_mix.log
for w in cfg2 cfg3; do
  PTX_BENCH_ROWS=$O/r3h_rows_$w.txt timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-x3 --no-autotune --no-cpu-baseline > $O/r3h_bench_$w.json 2> $O/r3h_bench_$w.err
  echo "$w exit $? $(tail -1 $O/r3h_bench_$w.json | cut -c1-200)"
done
