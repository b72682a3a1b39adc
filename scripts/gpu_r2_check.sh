#!/bin/bash
# round-2 inner loop: full GPU test-suite (or TESTS=-k expr), smoke, bench cfg2 (+ WORKLOADS="cfg3 cfg5" ...)
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
if [ -z "$NOTESTS" ]; then
timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -q -m gpu -p no:cacheprovider -x ${TESTS:+-k "$TESTS"} > gpurun_out/pytest.log 2>&1; echo "pytest exit $?"
grep -v "^tune\|amdgpu.ids" gpurun_out/pytest.log | tail -${TAIL:-15} | cut -c1-400
fi
if [ -z "$NOBENCH" ]; then
timeout 600 python bench.py --steps ${STEPS:-20} --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"
tail -1 gpurun_out/bench.log | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('VALUE', r['value'], r['unit'], r['ms_per_step'], 'ms', 'net frac', r['roofline_net']['frac'], 'parity', r['parity'])
print('DOM', r['roofline'])
for k,v in r['roofline_net']['per_kernel'].items(): print('  ', k, v)
" 2>&1 | cut -c1-300
tail -3 gpurun_out/bench.err
fi
for w in $WORKLOADS; do
timeout 900 python bench.py --workload $w --steps ${STEPS:-20} --warmup 5 > gpurun_out/bench_$w.log 2> gpurun_out/bench_$w.err; echo "bench $w exit $?"
tail -1 gpurun_out/bench_$w.log | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('VALUE', r['value'], r['unit'], r['ms_per_step'], 'ms', 'net frac', r['roofline_net']['frac'], 'conv_ms', r['roofline_net']['conv_ms_sum'], 'parity', r['parity'])
for k,v in r['roofline_net']['per_kernel'].items(): print('  ', k, v)
" 2>&1 | cut -c1-300
done
