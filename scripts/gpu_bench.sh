#!/bin/bash
# quick perf iteration: kernel unit tests (fast subset) + bench with the full tuning log
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
export PTX_TUNE_LOG=gpurun_out/tune_all.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider -x > gpurun_out/pytest_k.log 2>&1
echo "kernel tests exit $?"; tail -3 gpurun_out/pytest_k.log
timeout 600 python bench.py --steps ${STEPS:-10} --warmup 3 --verbose ${BENCH_ARGS} > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?"
grep "^tune" gpurun_out/bench.log | cut -c1-150
tail -1 gpurun_out/bench.log | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('VALUE', r['value'], 'clips/s', r['ms_per_step'], 'ms', 'net frac', r['roofline_net']['frac'], 'parity', r['parity'], 'cpu', r['cpu_baseline'] and r['cpu_baseline']['value'])
for k,v in r['roofline_net']['per_kernel'].items(): print('  ', k, v)
"
tail -3 gpurun_out/bench.err
