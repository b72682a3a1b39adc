#!/bin/bash
# round-6 GPU session 2: body kernel + chained tail (tests, probe), bench lines of configs 2 and 3 with the tuner's verdicts
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv_body" > gpurun_out/r06_s2_kernels.txt 2>&1; echo "kernel tests exit $?"; tail -5 gpurun_out/r06_s2_kernels.txt
python scripts/gpu_body_probe.py 8 > gpurun_out/r06_body_probe.txt 2>&1; echo "probe exit $?"
grep -v amdgpu.ids gpurun_out/r06_body_probe.txt | tail -14
PTX_TUNE_ITERS=4 PTX_TUNED_OUT=gpurun_out/r06_tuned_cfg2.json PTX_BENCH_ROWS=gpurun_out/r06_rows_cfg2_s2.txt timeout 900 python bench.py --steps 20 --warmup 5 --verbose > gpurun_out/r06_bench_cfg2_s2.log 2> gpurun_out/r06_bench_cfg2_s2.err; echo "bench cfg2 exit $?"
grep -h "body\|clip lanes" gpurun_out/r06_bench_cfg2_s2.log | head -30
python - <<'PY'
import json
for l in open("gpurun_out/r06_bench_cfg2_s2.log"):
    if l.startswith("{"):
        j = json.loads(l)
        print("cfg2:", j["value"], j["ms_per_step"], "lanes", j["config"]["clip_lanes"], "| other lanes leg:", j["clip_lanes"] and (j["clip_lanes"]["lanes"], j["clip_lanes"]["value"]), "| x3:", j["split_f16x3"] and j["split_f16x3"]["value"])
        print(j["launch_timing"])
        print({k: v for k, v in j["roofline_longest_launch"].items() if k in ("kernel", "frac", "avg_launch_ms", "frac_rocprof", "issued_frac")})
        print({k: v for k, v in j["roofline"].items() if k in ("kernel", "frac", "avg_launch_ms", "launches_per_step", "frac_rocprof")})
        print(j["parity"])
PY
cat gpurun_out/r06_rows_cfg2_s2.txt | grep -v "layer3\|layer4" | head -40
