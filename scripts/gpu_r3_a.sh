#!/bin/bash
# Round-3 payload A: GPU tests, then A/B of the chained convs (PTX_CHAIN=1 / 0) on config 2 and config 3.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
echo "== pytest gpu"
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu -p no:cacheprovider -x ${PYTEST_ARGS} > $O/r3a_pytest.log 2>&1
echo "pytest exit $?" | tee -a $O/r3a_pytest.log
tail -30 $O/r3a_pytest.log
for chain in 1 0; do
  echo "== bench cfg2 chain=$chain"
  PTX_CHAIN=$chain PTX_BENCH_ROWS=$O/r3a_rows_cfg2_c$chain.txt PTX_TUNED_OUT=$O/r3a_tuned_cfg2_c$chain.json \
    timeout 600 python bench.py --steps 20 --warmup 5 --no-x3 $( [ $chain = 0 ] && echo --no-cpu-baseline ) > $O/r3a_bench_cfg2_c$chain.json 2> $O/r3a_bench_cfg2_c$chain.err
  echo "exit $?"; tail -c 600 $O/r3a_bench_cfg2_c$chain.json | head -c 600; echo
done
for chain in 1 0; do
  echo "== bench cfg3 chain=$chain"
  PTX_CHAIN=$chain PTX_BENCH_ROWS=$O/r3a_rows_cfg3_c$chain.txt PTX_TUNED_OUT=$O/r3a_tuned_cfg3_c$chain.json \
    timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-x3 $( [ $chain = 0 ] && echo --no-cpu-baseline ) > $O/r3a_bench_cfg3_c$chain.json 2> $O/r3a_bench_cfg3_c$chain.err
  echo "exit $?"; tail -c 600 $O/r3a_bench_cfg3_c$chain.json | head -c 600; echo
done
grep -h '"value"' $O/r3a_bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d['roofline_net']['frac'], d.get('parity'))
    except Exception as e: print('bad line', e)
"
