#!/bin/bash
export PYTHONDONTWRITEBYTECODE=1
for a in 0 1 2 3 4 7; do echo "=== PTX_ABLATE=$a"; PTX_ABLATE=$a python scripts/conv_micro.py $1 2>&1 | grep -v amdgpu.ids; done
