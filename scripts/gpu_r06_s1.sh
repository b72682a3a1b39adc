#!/bin/bash
# round-6 GPU session 1: body-kernel probe + its kernel tests, the new model-level tests, one bench line of config 2
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
python scripts/gpu_body_probe.py 8 > gpurun_out/r06_body_probe.txt 2>&1; echo "probe exit $?"
tail -12 gpurun_out/r06_body_probe.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv_body" > gpurun_out/r06_s1_kernels.txt 2>&1; echo "kernel tests exit $?"; tail -5 gpurun_out/r06_s1_kernels.txt
timeout 1500 python -m pytest tests/test_gpu_models.py -x -q -s -k "bench_line or strong_scaling or adversarial" > gpurun_out/r06_s1_models.txt 2>&1; echo "model tests exit $?"; tail -12 gpurun_out/r06_s1_models.txt
PTX_TUNED_OUT=gpurun_out/r06_tuned_s1.json PTX_BENCH_ROWS=gpurun_out/r06_rows_cfg2_s1.txt timeout 900 python bench.py --steps 20 --warmup 5 --verbose > gpurun_out/r06_bench_cfg2_s1.log 2> gpurun_out/r06_bench_cfg2_s1.err; echo "bench exit $?"
tail -c 3000 gpurun_out/r06_bench_cfg2_s1.log | head -c 3000
grep -h "body\|clip lanes" gpurun_out/r06_bench_cfg2_s1.log | head -20
