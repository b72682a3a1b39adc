#!/bin/bash
# kw-reuse chained tiles (split operands): chain tests + config-2 bench that tunes the missing chain / alt entries
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 PTX_TUNE_ITERS=8
O=gpurun_out
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -k "chain" > $O/r3m_pytest.log 2>&1; echo "pytest exit $?"; tail -3 $O/r3m_pytest.log
PTX_TUNE_VERBOSE=1 PTX_BENCH_ROWS=$O/r3m_rows_cfg2.txt PTX_TUNED_OUT=$O/r3m_tuned_cfg2.json timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r3m_bench_cfg2.json 2> $O/r3m_bench_cfg2.err
echo "cfg2 exit $?"; python - <<'E'
import json
d=json.loads(open('gpurun_out/r3m_bench_cfg2.json').read().strip().splitlines()[-1])
print(d['value'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'])
E
grep "chain" $O/r3m_rows_cfg2.txt.x3 | cut -c1-150
grep -i "chain" $O/r3m_bench_cfg2.err | cut -c1-200 | tail -12
PTX_CHAIN_FORCE=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-autotune > $O/r3m_bench_cfg2_force.json 2> /dev/null
python - <<'E'
import json
d=json.loads(open('gpurun_out/r3m_bench_cfg2_force.json').read().strip().splitlines()[-1])
print('forced chain:', d['value'], d['split_f16x3']['value'], d['split_f16x3']['ms_per_step'])
E
