#!/bin/bash
# Round-3 payload C: GPU tests; key-split attention A/B on config 3; table-driven (no autotune) runs of config 2 / 3.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
echo "== pytest gpu"
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu -p no:cacheprovider ${PYTEST_ARGS} > $O/r3c_pytest.log 2>&1
echo "pytest exit $?" | tee -a $O/r3c_pytest.log
tail -8 $O/r3c_pytest.log
for ks in 0 1; do
  echo "== bench cfg3 ksplit=$ks (tuned table, no autotune)"
  PTX_NL_KSPLIT=$ks PTX_BENCH_ROWS=$O/r3c_rows_cfg3_ks$ks.txt timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-x3 --no-autotune \
    $( [ $ks = 0 ] && echo --no-cpu-baseline ) > $O/r3c_bench_cfg3_ks$ks.json 2> $O/r3c_bench_cfg3_ks$ks.err
  echo "exit $?"; grep nonlocal_attention $O/r3c_rows_cfg3_ks$ks.txt
done
echo "== bench cfg2 (tuned table, no autotune)"
timeout 600 python bench.py --steps 20 --warmup 5 --no-x3 --no-autotune --no-cpu-baseline > $O/r3c_bench_cfg2.json 2> $O/r3c_bench_cfg2.err; echo "exit $?"
grep -h '"value"' $O/r3c_bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d['roofline_net']['frac'], d.get('parity'))
    except Exception as e: print('bad line', e)
"
