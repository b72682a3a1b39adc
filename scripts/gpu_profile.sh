#!/bin/bash
# rocprofv3 evidence for profiles/: (1) kernel trace + stats of the bench command, (2) PMC passes.
export PYTHONDONTWRITEBYTECODE=1
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-x3 ${BENCH_ARGS}"
echo "== kernel trace"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1
echo "trace exit $?"
find $OUT/trace -name "*stats*" | head
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  echo "== pmc $tag"
  timeout 600 rocprofv3 --pmc $pmc --kernel-trace -d $OUT/pmc_$tag -o bench -- $CMD --no-autotune > $OUT/pmc_$tag.log 2>&1
  echo "pmc exit $?"
done
cd $REPO
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
tail -60 $OUT/summary.txt
du -sh $OUT
# keep only small artefacts (<64 MiB merge limit)
find $OUT -name "*.db" -size +20M -delete
