#!/bin/bash
# Functional check of bench.py's N > 1 branches on a 1-GPU box: 2 ranks sharing the device, collectives over gloo
# (PTX_BENCH_BACKEND=gloo).  NOT a scaling figure -- it shows the tuned-table broadcast, the sharding (weak / strong, config 2 /
# config 4), the all-gather of logits and the self-check (`distributed_check`) executing with real GPU forwards.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1 PTX_BENCH_BACKEND=gloo
O=gpurun_out
run() {  # name, extra args
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29540 + RANDOM % 50)) \
    bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-x3 $2 > $O/gloo2_$1.json 2> $O/gloo2_$1.err
  echo "$1 exit $?"
  N=$1 python - <<'E'
import json, os
d = json.loads(open("gpurun_out/gloo2_%s.json" % os.environ["N"]).read().strip().splitlines()[-1])
print(d["value"], d["unit"], d["scaling"], d["config"], d["distributed_check"], {k: d["ranks_seen"][k] for k in ("world_size", "device_count", "distinct_devices", "tuned_entries_broadcast")})
E
}
run cfg2_weak ""
run cfg2_strong "--scaling strong"
run cfg4_weak "--workload cfg4"
run cfg4_strong "--workload cfg4 --scaling strong"
