#!/bin/bash
# x3 tiles with the row-major epilogue: tests + careful re-tune of the split-operand legs (config 2 / 3 benches tune both legs)
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 PTX_TUNE_ITERS=6
O=gpurun_out
timeout 600 python -m pytest tests -q -m gpu -p no:cacheprovider -k "x3 or split" > $O/r3i_pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/r3i_pytest.log
PTX_BENCH_ROWS=$O/r3i_rows_cfg2.txt PTX_TUNED_OUT=$O/r3i_tuned_cfg2.json timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r3i_bench_cfg2.json 2> $O/r3i_bench_cfg2.err
echo "cfg2 exit $?"; python - <<'E'
import json
d=json.loads(open('gpurun_out/r3i_bench_cfg2.json').read().strip().splitlines()[-1])
print(d['value'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'], d['split_f16x3']['roofline_net'])
E
PTX_FULL_TUNE=1 PTX_TUNED_OUT=$O/r3i_tuned_cfg3.json timeout 900 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-cpu-baseline > $O/r3i_bench_cfg3.json 2> $O/r3i_bench_cfg3.err
echo "cfg3 exit $?"; python - <<'E'
import json
d=json.loads(open('gpurun_out/r3i_bench_cfg3.json').read().strip().splitlines()[-1])
print(d['value'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'])
E
PTX_FULL_TUNE=1 PTX_TUNED_OUT=$O/r3i_tuned_cfg4.json timeout 900 python bench.py --workload cfg4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r3i_bench_cfg4.json 2> $O/r3i_bench_cfg4.err; echo "cfg4 exit $?"
PTX_PRECISION=x3 PTX_RETUNE=1 ZOO_NO_CPU=1 PTX_TUNED_OUT=$O/r3i_tuned_zoo_x3.json timeout 1500 python scripts/gpu_zoo_bench.py > $O/r3i_zoo_x3.log 2>&1; echo "zoo x3 exit $?"; tail -17 $O/r3i_zoo_x3.log | cut -c1-150
cp $O/zoo_bench.json $O/r3i_zoo_bench_x3.json
