#!/bin/bash
# round-6 GPU session 5: stem column-tile loop (VERDICT r5 #2c) -- kernel + model tests, cfg3 A/B (PTX_STEM_F32_NTLOOP=0|1), cfg2 check
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "stem" > gpurun_out/r06_s5_kernels.txt 2>&1; echo "kernel tests exit $?"; tail -3 gpurun_out/r06_s5_kernels.txt
timeout 1200 python -m pytest tests/test_gpu_models.py -x -q -k "r2plus1d or cfg3 or composite or nl" > gpurun_out/r06_s5_models.txt 2>&1; echo "model tests exit $?"; tail -3 gpurun_out/r06_s5_models.txt
for v in 1 0 1 0; do
  PTX_STEM_F32_NTLOOP=$v PTX_BENCH_ROWS=gpurun_out/r06_rows_cfg3_ntloop$v.txt timeout 600 python bench.py --workload cfg3 --steps 20 --warmup 5 --no-x3 --no-lanes --no-cpu-baseline > gpurun_out/r06_bench_cfg3_ntloop$v.log 2> gpurun_out/r06_bench_cfg3_ntloop$v.err; echo "bench cfg3 ntloop=$v exit $?"
  python - <<PY
import json
for l in open("gpurun_out/r06_bench_cfg3_ntloop$v.log"):
    if l.startswith("{"):
        j = json.loads(l); print("cfg3 ntloop=$v:", j["value"], j["ms_per_step"])
PY
  grep "conv1.spatial\|conv1.temporal" gpurun_out/r06_rows_cfg3_ntloop$v.txt
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-x3 --no-lanes --no-cpu-baseline > gpurun_out/r06_bench_cfg2_s5.log 2> gpurun_out/r06_bench_cfg2_s5.err; echo "bench cfg2 exit $?"
python - <<'PY'
import json
for l in open("gpurun_out/r06_bench_cfg2_s5.log"):
    if l.startswith("{"):
        j = json.loads(l); print("cfg2:", j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"] if "roofline" in j else None)
PY
