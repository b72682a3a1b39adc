#!/bin/bash
# round-6 GPU session 3: body kernel probe (balanced case, improved chained tail), kernel tests, bench cfg2
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv_body" > gpurun_out/r06_s3_kernels.txt 2>&1; echo "kernel tests exit $?"; tail -4 gpurun_out/r06_s3_kernels.txt
python scripts/gpu_body_probe.py 8 > gpurun_out/r06_body_probe.txt 2>&1; echo "probe exit $?"
grep -v amdgpu.ids gpurun_out/r06_body_probe.txt | tail -18
PTX_TUNE_ITERS=4 PTX_TUNED_OUT=gpurun_out/r06_tuned_cfg2.json PTX_BENCH_ROWS=gpurun_out/r06_rows_cfg2_s3.txt timeout 900 python bench.py --steps 20 --warmup 5 --verbose --no-x3 > gpurun_out/r06_bench_cfg2_s3.log 2> gpurun_out/r06_bench_cfg2_s3.err; echo "bench cfg2 exit $?"
grep -h "body\|clip lanes" gpurun_out/r06_bench_cfg2_s3.log | head -12
python - <<'PY'
import json
for l in open("gpurun_out/r06_bench_cfg2_s3.log"):
    if l.startswith("{"):
        j = json.loads(l)
        print("cfg2:", j["value"], j["ms_per_step"], "lanes", j["config"]["clip_lanes"], "| other lanes leg:", j["clip_lanes"] and (j["clip_lanes"]["lanes"], j["clip_lanes"]["value"]))
        print(j["launch_timing"])
        print({k: v for k, v in j["roofline_longest_launch"].items() if k in ("kernel", "frac", "avg_launch_ms", "frac_rocprof", "issued_frac")})
PY
grep "layer1" gpurun_out/r06_rows_cfg2_s3.txt
