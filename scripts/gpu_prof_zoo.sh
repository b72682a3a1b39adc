#!/bin/bash
# rocprofv3 kernel trace of one zoo-bench case: bash scripts/gpu_prof_zoo.sh '<case substring>' <tag>
export PYTHONDONTWRITEBYTECODE=1 ZOO_NO_CPU=1
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$2; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o zoo -- python $REPO/scripts/gpu_zoo_bench.py "$1" > $OUT/trace.log 2>&1; echo "trace exit $?"
cd $REPO
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
head -40 $OUT/summary.txt | cut -c1-160
