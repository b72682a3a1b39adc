#!/bin/bash
# round-end check without the rocprof passes: full GPU suite, smoke, bench.py for every BASELINE configuration
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 gpurun_out/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-330
for w in cfg1 cfg3 cfg4 cfg5; do
  timeout 500 python bench.py --workload $w --steps 10 --warmup 3 > gpurun_out/bench_$w.log 2> gpurun_out/bench_$w.err; echo "$w exit $?"; tail -1 gpurun_out/bench_$w.log | cut -c1-200
done
