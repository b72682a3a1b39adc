#!/bin/bash
# round-2 iteration helper: a few bench runs with different environments, each printing one summary line.
#   RUNS="name:ENV=val,ENV2=val:workload ..."  e.g. RUNS="base::cfg2 stem32:PTX_STEM_LD=32:cfg2"
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
for r in $RUNS; do
  name=$(echo $r | cut -d: -f1); envs=$(echo $r | cut -d: -f2 | tr ',' ' '); w=$(echo $r | cut -d: -f3)
  ( for e in $envs; do export $e; done
    export PTX_BENCH_ROWS=gpurun_out/rows_$name.txt
    timeout 600 python bench.py --workload ${w:-cfg2} --steps ${STEPS:-20} --warmup 5 ${BENCH_ARGS} > gpurun_out/bench_$name.log 2> gpurun_out/bench_$name.err
    echo "== $name ($envs, ${w:-cfg2}) exit $?" )
  tail -1 gpurun_out/bench_$name.log | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('VALUE', r['value'], r['unit'], r['ms_per_step'], 'ms', 'net', r['roofline_net']['frac'], 'conv_ms', r['roofline_net']['conv_ms_sum'], 'nonconv', r.get('non_conv_ms'), 'parity', r['parity'] and r['parity']['max_abs_dlogits'])
print('DOM', r['roofline']['kernel'], r['roofline']['achieved'], r['roofline']['avg_launch_ms'])
for k,v in r.get('roofline_hbm',{}).items(): print('  HBM', k, v['launches'], v['ms'], v['achieved'], v['frac'])
" 2>&1 | cut -c1-250
  tail -2 gpurun_out/bench_$name.err | cut -c1-300
done
