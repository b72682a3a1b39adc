#!/bin/bash
# Diagnostic build: libptx_amd with the phase clocks of the 3x3 generator-stage kernel (-DPTX_C3_TIMELINE), of the generic
# implicit-GEMM kernel (-DPTX_IGEMM_TIMELINE) and of the direct fp32 stem (-DPTX_STEM_TIMELINE) compiled in, linked with the product's other objects, written NEXT TO this script
# (scripts/micro/libptx_amd_tl.so: git-ignored, travels with a gpurun snapshot).  The product library is not touched.
# Used by scripts/gpu_c3_timeline.py and scripts/gpu_igemm_timeline.py.  (conv_igemm.hip takes ~4 minutes.)
set -e
cd "$(dirname "$0")/../.."
python pretorched-x_amd/csrc/build.py > /dev/null
CS=pretorched-x_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F -DPTX_C3_TIMELINE -c $CS/gen_stage_f16.hip -o scripts/micro/gen_stage_f16_tl.o &
/opt/rocm/bin/hipcc $F -DPTX_STEM_TIMELINE -c $CS/conv_stem_f32.hip -o scripts/micro/conv_stem_f32_tl.o &
if [ ! -f scripts/micro/conv_igemm_tl.o ] || [ $CS/conv_igemm.hip -nt scripts/micro/conv_igemm_tl.o ] || [ $CS/conv_igemm_kernel.h -nt scripts/micro/conv_igemm_tl.o ]; then
  /opt/rocm/bin/hipcc $F -DPTX_IGEMM_TIMELINE -c $CS/conv_igemm.hip -o scripts/micro/conv_igemm_tl.o &
fi
wait
OBJS=$(ls $CS/*.o | grep -v "gen_stage_f16.o\|conv_igemm.o\|conv_stem_f32.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/micro/libptx_amd_tl.so $OBJS scripts/micro/gen_stage_f16_tl.o scripts/micro/conv_igemm_tl.o scripts/micro/conv_stem_f32_tl.o
echo built scripts/micro/libptx_amd_tl.so
