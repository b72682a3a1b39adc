#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider -s > gpurun_out/pytest.log 2>&1; echo "pytest exit $?"; grep -E "max\|dlogits\||passed|failed" gpurun_out/pytest.log | tail -12
timeout 600 python scripts/gpu_cfg3.py > gpurun_out/cfg3.log 2>&1; echo "cfg3 exit $?"; grep -v amdgpu.ids gpurun_out/cfg3.log | tail -30
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_torchrun.log 2>&1; echo "torchrun bench exit $?"; tail -1 gpurun_out/bench_torchrun.log | cut -c1-300
