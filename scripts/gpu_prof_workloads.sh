#!/bin/bash
# rocprofv3 kernel traces of bench.py for the non-headline BASELINE configurations (tile choices come from the
# tuned table, so the trace holds steady-state launches only)
export PYTHONDONTWRITEBYTECODE=1
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-cfg3 cfg4 cfg5}; do
  OUT=$REPO/gpurun_out/prof_$w; rm -rf $OUT; mkdir -p $OUT
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python $REPO/bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1; echo "$w trace exit $?"
  python $REPO/scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
  find $OUT -name "*.db" -delete
  head -14 $OUT/summary.txt | cut -c1-150
done
