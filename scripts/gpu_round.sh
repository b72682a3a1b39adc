#!/bin/bash
# One gpurun payload: GPU tests, then a short bench (logs merged back under gpurun_out/).
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
echo "== rocminfo" > gpurun_out/env.log
(rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -8; nproc; free -g | head -2) >> gpurun_out/env.log 2>&1
echo "== pytest gpu"
timeout ${PYTEST_TIMEOUT:-900} python -m pytest tests -q -m gpu -p no:cacheprovider ${PYTEST_ARGS} > gpurun_out/pytest.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest.log
tail -40 gpurun_out/pytest.log
echo "== bench"
timeout ${BENCH_TIMEOUT:-600} python bench.py --steps ${STEPS:-10} --warmup 3 --verbose > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?" | tee -a gpurun_out/bench.err
tail -45 gpurun_out/bench.log; tail -5 gpurun_out/bench.err
