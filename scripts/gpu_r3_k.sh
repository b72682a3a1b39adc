#!/bin/bash
# split-operand chained tiles (conv -> 1x1x1 conv, both GEMMs x3): tests + re-tune of the x3 legs of configs 2 / 3
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 PTX_TUNE_ITERS=6
O=gpurun_out
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "chain or x3 or split" > $O/r3k_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/r3k_pytest.log
PTX_BENCH_ROWS=$O/r3k_rows_cfg2.txt PTX_TUNED_OUT=$O/r3k_tuned_cfg2.json timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r3k_bench_cfg2.json 2> $O/r3k_bench_cfg2.err
echo "cfg2 exit $?"; python - <<'E'
import json
d=json.loads(open('gpurun_out/r3k_bench_cfg2.json').read().strip().splitlines()[-1])
print(d['value'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'], d['split_f16x3'].get('max_abs_dlogits'))
E
grep "chain\|tune.*pair" $O/r3k_rows_cfg2.txt.x3 | cut -c1-150
grep "chain .* pair" $O/r3k_bench_cfg2.err | cut -c1-150 | tail -8
PTX_BENCH_ROWS=$O/r3k_rows_cfg3.txt PTX_FULL_TUNE=1 PTX_TUNED_OUT=$O/r3k_tuned_cfg3.json timeout 900 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline > $O/r3k_bench_cfg3.json 2> $O/r3k_bench_cfg3.err
echo "cfg3 exit $?"; python - <<'E'
import json
d=json.loads(open('gpurun_out/r3k_bench_cfg3.json').read().strip().splitlines()[-1])
print(d['value'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'], d['split_f16x3'].get('max_abs_dlogits'))
E
grep "chain" $O/r3k_rows_cfg3.txt.x3 | cut -c1-150
grep "chain .* pair" $O/r3k_bench_cfg3.err | cut -c1-150 | tail -24
