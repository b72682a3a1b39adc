#!/bin/bash
# A/B an env switch on the bench (no CPU baseline): usage gpu_ab.sh VAR val_a val_b
export PYTHONDONTWRITEBYTECODE=1
for v in $2 $3 $2 $3; do
  echo "== $1=$v"; env $1=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline']['kernel'], r['roofline']['achieved'], r['roofline']['avg_launch_ms'])"
done
