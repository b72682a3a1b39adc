#!/bin/bash
# rocprofv3 kernel trace + separate PMC passes (MFMA / LDS / FETCH / WRITE) of bench.py for one workload.
#   W=cfg5 [ENVS="PTX_PRECISION=x3"] bash scripts/gpu_prof_pmc.sh   ->  gpurun_out/prof_$W$TAG/summary.txt
#   W=stem PROF_CMD="python scripts/gpu_stem_scale_probe.py 8" PASSES="0 1" bash scripts/gpu_prof_pmc.sh    (a probe script)
export PYTHONDONTWRITEBYTECODE=1
for e in $ENVS; do export $e; done
REPO=$(pwd)
W=${W:-cfg5}
OUT=$REPO/gpurun_out/prof_$W$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --workload $W --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-x3 --no-lanes --lanes 1 ${BENCH_ARGS}"   # (single plan: under clip lanes two kernels share the chip)
NOTUNE="--no-autotune"
if [ -n "$PROF_CMD" ]; then CMD="$PROF_CMD"; NOTUNE=""; fi      # any other command (a kernel probe script): W / TAG only name the output
# the trace pass runs with the tuned table too (--no-autotune): tuner launches would pollute the per-kernel averages
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD $NOTUNE > $OUT/trace.log 2>&1; echo "trace exit $?"
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi      # PASSES="1 3 4": only those counter groups
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pmc --kernel-trace -d $OUT/pmc_$tag -o bench -- $CMD $NOTUNE > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag exit $?"
done
cd $REPO
python scripts/summarize_prof.py $OUT ${LASTN:-} > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
du -sh $OUT
