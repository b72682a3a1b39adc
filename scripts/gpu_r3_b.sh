#!/bin/bash
# Round-3 payload B: GPU tests, full-tune benches of config 2 / 3 with the row-major-epilogue tiles, 144-wide tiles and
# the measured chain-vs-pair decision; BigGAN chunk probe.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
echo "== pytest gpu"
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu -p no:cacheprovider ${PYTEST_ARGS} > $O/r3b_pytest.log 2>&1
echo "pytest exit $?" | tee -a $O/r3b_pytest.log
tail -25 $O/r3b_pytest.log
echo "== bench cfg2 (full tune)"
PTX_TUNE_LOG=$O/r3b_tune_cfg2.log PTX_BENCH_ROWS=$O/r3b_rows_cfg2.txt PTX_TUNED_OUT=$O/r3b_tuned_cfg2.json \
  timeout 900 python bench.py --steps 20 --warmup 5 --no-x3 > $O/r3b_bench_cfg2.json 2> $O/r3b_bench_cfg2.err
echo "exit $?"; tail -3 $O/r3b_bench_cfg2.err
echo "== bench cfg3 (full tune)"
PTX_FULL_TUNE=1 PTX_TUNE_LOG=$O/r3b_tune_cfg3.log PTX_BENCH_ROWS=$O/r3b_rows_cfg3.txt PTX_TUNED_OUT=$O/r3b_tuned_cfg3.json \
  timeout 900 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-x3 > $O/r3b_bench_cfg3.json 2> $O/r3b_bench_cfg3.err
echo "exit $?"; tail -3 $O/r3b_bench_cfg3.err
echo "== biggan chunk probe"
timeout 300 python scripts/gpu_biggan_chunk_probe.py fp16 > $O/r3b_biggan_probe.log 2>&1; tail -12 $O/r3b_biggan_probe.log
grep -h '"value"' $O/r3b_bench_*.json | python -c "
import sys, json
for l in sys.stdin:
    try:
        d = json.loads(l); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d['roofline_net']['frac'], d.get('parity'))
    except Exception as e: print('bad line', e)
"
grep -h "chain\|pair" $O/r3b_tune_cfg2.log | tail -12
