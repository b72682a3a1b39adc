#!/bin/bash
# split-operand stem on 32-row wave tiles, two workgroups per CU: tests + the config-2 / 3 / 4 benches (both legs, table-driven)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "stem or x3 or split" > $O/r3s_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/r3s_pytest.log
for w in cfg2 cfg3 cfg4; do
PTX_BENCH_ROWS=$O/r3s_rows_$w.txt timeout 900 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-autotune > $O/r3s_bench_$w.json 2> $O/r3s_bench_$w.err
echo "$w exit $?"; W=$w python - <<'E'
import json, os
d=json.loads(open('gpurun_out/r3s_bench_%s.json' % os.environ['W']).read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'])
E
grep "^conv1\|stem " $O/r3s_rows_$w.txt.x3 | cut -c1-140
done
