#!/bin/bash
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
timeout 300 python scripts/gpu_two_stream_probe.py resnet3d50 8x3x16x224x224 2 > $O/r3f_two_stream_cfg2.log 2>&1; tail -5 $O/r3f_two_stream_cfg2.log
timeout 300 python scripts/gpu_two_stream_probe.py resnet3d50 8x3x16x224x224 4 > $O/r3f_four_stream_cfg2.log 2>&1; tail -4 $O/r3f_four_stream_cfg2.log
timeout 300 python scripts/gpu_two_stream_probe.py nonlocal_r2plus1d50 8x3x32x112x112 2 > $O/r3f_two_stream_cfg3.log 2>&1; tail -4 $O/r3f_two_stream_cfg3.log
