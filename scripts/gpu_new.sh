#!/bin/bash
# validation of newly added paths: kernel tests, selected model tests (TESTS env = -k expression), then the bench
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -p no:cacheprovider > gpurun_out/pytest_k.log 2>&1
echo "kernel tests exit $?"; tail -25 gpurun_out/pytest_k.log | cut -c1-300
timeout 1500 python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -s -k "${TESTS:-trn or slowfast or frames or small}" > gpurun_out/pytest_m.log 2>&1
echo "model tests exit $?"; grep -v "^tune\|amdgpu.ids" gpurun_out/pytest_m.log | tail -40 | cut -c1-300
if [ -z "$NOBENCH" ]; then
timeout 600 python bench.py --steps ${STEPS:-10} --warmup 3 > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench exit $?"
tail -1 gpurun_out/bench.log | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('VALUE', r['value'], 'clips/s', r['ms_per_step'], 'ms', 'net frac', r['roofline_net']['frac'], 'parity', r['parity'])
"
fi
