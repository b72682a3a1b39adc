#!/bin/bash
# conv-program probe: one process per case, each under its own short timeout (a wedged launch costs seconds, not the call)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_probe_${1:-a}
mkdir -p $OUT
export PTX_PROG_SPIN_LIMIT=${PTX_PROG_SPIN_LIMIT:-200000}
for c in ${CASES:-plain one two split block full layer3 layer4}; do
  timeout 120 python scripts/gpu_prog_probe.py $c > $OUT/$c.log 2>&1
  echo "case $c rc=$?" | tee -a $OUT/summary.txt
  grep -v amdgpu.ids $OUT/$c.log | tail -${TAILN:-14}
done
