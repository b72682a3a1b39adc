#!/bin/bash
# split-operand chained tiles, second pass: the full GPU suite + table-driven benches of configs 2 / 3
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/r3l_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/r3l_pytest.log
for w in cfg2 cfg3; do
PTX_BENCH_ROWS=$O/r3l_rows_$w.txt timeout 900 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-autotune > $O/r3l_bench_$w.json 2> $O/r3l_bench_$w.err
echo "$w exit $?"; W=$w python - <<'E'
import json, os
d=json.loads(open('gpurun_out/r3l_bench_%s.json' % os.environ['W']).read().strip().splitlines()[-1])
print(d['value'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'])
E
done
for w in cfg2 cfg3; do
PTX_CHAIN=0 timeout 900 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-autotune > $O/r3l_bench_${w}_nochain.json 2> /dev/null
echo "$w PTX_CHAIN=0 exit $?"; W=$w python - <<'E'
import json, os
d=json.loads(open('gpurun_out/r3l_bench_%s_nochain.json' % os.environ['W']).read().strip().splitlines()[-1])
print(d['value'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'])
E
done
