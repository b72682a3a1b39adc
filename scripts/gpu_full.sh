#!/bin/bash
# full GPU test-suite + smoke + bench + rocprofv3 evidence (kernel trace / stats + PMC passes)
export PYTHONDONTWRITEBYTECODE=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/pytest.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit $?"; tail -2 gpurun_out/smoke.log
export PTX_TUNED_OUT=gpurun_out/tuned_gfx950.json
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; tail -1 gpurun_out/bench.log | cut -c1-400
cp gpurun_out/tuned_gfx950.json pretorched-x_amd/tuned_gfx950.json 2>/dev/null
REPO=$(pwd); OUT=$REPO/gpurun_out/prof; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-autotune"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- $CMD > $OUT/trace.log 2>&1; echo "trace exit $?"
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $pmc --kernel-trace -d $OUT/pmc_$tag -o bench -- $CMD > $OUT/pmc_$tag.log 2>&1; echo "pmc $tag exit $?"
done
cd $REPO
python scripts/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
find $OUT -name "*.db" -delete
tail -30 $OUT/summary.txt | cut -c1-200
