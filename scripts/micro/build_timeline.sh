#!/bin/bash
# Diagnostic build: libptx_amd with the 3x3 generator-stage kernel's phase clock compiled in (-DPTX_C3_TIMELINE), linked from the
# product's own objects, written NEXT TO this script (scripts/micro/libptx_amd_tl.so: git-ignored, travels with a gpurun
# snapshot).  The product library is not touched.  Used by scripts/gpu_c3_timeline.py.
set -e
cd "$(dirname "$0")/../.."
python pretorched-x_amd/csrc/build.py > /dev/null
CS=pretorched-x_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -DPTX_C3_TIMELINE \
    -c $CS/gen_stage_f16.hip -o scripts/micro/gen_stage_f16_tl.o
OBJS=$(ls $CS/*.o | grep -v gen_stage_f16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/libptx_amd_tl.so $OBJS scripts/micro/gen_stage_f16_tl.o
echo built scripts/micro/libptx_amd_tl.so
