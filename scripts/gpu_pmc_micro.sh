#!/bin/bash
# PMC counters for conv_micro shapes: usage: gpu_pmc_micro.sh "<conv_micro args>"
export PYTHONDONTWRITEBYTECODE=1
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_micro; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  (cd $REPO && timeout 300 rocprofv3 --pmc $pmc --kernel-trace -d $OUT/p$i -o m -- python scripts/conv_micro.py $1 > $OUT/p$i.log 2>&1)
  echo "pmc set $i exit $?"
done
cd $REPO
python - <<'PY'
import glob, sqlite3
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_micro/p*/*.db")):
    db = sqlite3.connect(f)
    for k, c, v, dur, gx in db.execute("select kernel_name, counter_name, value, duration, grid_size_x from counters_collection"):
        if "conv_igemm" not in k:
            continue
        key = k.replace("void ptx::conv_igemm_kernel", "").replace("(ptx::ConvArgs)", "") + " grid=%d" % gx
        acc[key][c].append(v)
        acc[key]["dur_us"].append(dur / 1e3)
for k, v in acc.items():
    print("==", k)
    for c in sorted(v):
        vals = v[c]
        print("   %-28s %14.5g" % (c, sum(vals) / len(vals)))
PY
find $OUT -name "*.db" -delete
