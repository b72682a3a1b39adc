#!/bin/bash
# PMC traffic tables of configs 4 and 5 (rocprofv3 trace + SQ MFMA / FETCH_SIZE / WRITE_SIZE passes, every pass --no-autotune),
# then the bench lines of configs 3 / 4 / 5 replaying the table of their OWN workload in roofline.traffic.
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05_traffic
mkdir -p $O
export PTX_COMMIT=$(cat .commit_for_gpurun 2>/dev/null || echo unknown)
for w in cfg4 cfg5; do
  tag=_fp32; [ $w = cfg5 ] && tag=_f16
  W=$w TAG=$tag STEPS=10 PASSES="1 3 4" bash scripts/gpu_prof_pmc.sh 2>&1 | tail -5 | tee -a $O/summary.txt
  cp gpurun_out/prof_$w$tag/summary.txt $O/rocprofv3_${w}${tag}_summary.txt
  python scripts/pmc_traffic_json.py gpurun_out/prof_$w$tag/summary.txt profiles/r05_pmc_traffic_$w.json "$PTX_COMMIT" \
    "python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-x3 --no-autotune" | tee -a $O/summary.txt
  cp profiles/r05_pmc_traffic_$w.json $O/
done
for w in cfg3 cfg4 cfg5; do
  PTX_BENCH_ROWS=$O/rows_$w.txt timeout 700 python bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?" | tee -a $O/summary.txt
  python - $O/bench_$w.json <<'PY' | tee -a $O/summary.txt
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]
    print(d["config"]["workload"][:40], d["value"], d["unit"], "| dominant", r["kernel"], "achieved", r["achieved"], "traffic", r["traffic"],
          "source", (r.get("traffic_source") or {}).get("file"))
except Exception as e:
    print("parse failed", e)
PY
done
