#!/bin/bash
# Round-6 evidence: full GPU suite, smoke, one bench line per BASELINE configuration (+ rows), rocprofv3 trace + PMC passes of
# configs 2 and 3, the N > 1 branches of bench.py on 2 ranks sharing the GPU over gloo (functional, --scaling both).
cd "$(dirname "$0")/.."
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r06_final
mkdir -p $O
export PTX_COMMIT=$(cat .commit_for_gpurun 2>/dev/null || echo unknown)
(rocminfo | grep -E "Marketing|gfx|Compute Unit" | head -6; nproc; free -g | head -2) > $O/env.log 2>&1
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest exit $?" | tee -a $O/summary.txt; tail -3 $O/pytest.log | tee -a $O/summary.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke exit $?" | tee -a $O/summary.txt; tail -1 $O/smoke.log | tee -a $O/summary.txt
fi
PTX_BENCH_ROWS=$O/rows_cfg2.txt timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_cfg2.json 2> $O/bench_cfg2.err; echo "bench cfg2 exit $?" | tee -a $O/summary.txt; tail -1 $O/bench_cfg2.json | cut -c1-260 | tee -a $O/summary.txt
for w in cfg1 cfg3 cfg4 cfg5; do
  PTX_BENCH_ROWS=$O/rows_$w.txt timeout 700 python bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w exit $?" | tee -a $O/summary.txt; tail -1 $O/bench_$w.json | cut -c1-220 | tee -a $O/summary.txt
done
PTX_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-x3 --no-autotune --scaling both > $O/gloo2_cfg2_both.json 2> $O/gloo2_cfg2_both.err; echo "gloo2 both exit $?" | tee -a $O/summary.txt
python - <<'E' | tee -a $O/summary.txt
import json
try:
    for l in open("gpurun_out/r06_final/gloo2_cfg2_both.json"):
        if l.startswith("{"):
            d = json.loads(l)
            print(d["scaling"], d["value"], d["unit"], d["rank_ms_per_step"], d["distributed_check"], {k: d["ranks_seen"][k] for k in ("world_size", "distinct_devices")})
except Exception as e:
    print("gloo parse failed", e)
E
timeout 300 python scripts/gpu_r05_lane_rows.py 2>/dev/null | tee $O/lane_rows.txt | tail -9 | tee -a $O/summary.txt
if [ -z "$SKIP_PROF" ]; then
W=cfg2 ENVS="PTX_PROGRAM=0" TAG=_fp32 STEPS=15 bash scripts/gpu_prof_pmc.sh 2>&1 | tail -7 | tee -a $O/summary.txt
W=cfg3 ENVS="PTX_PROGRAM=0" TAG=_fp32 STEPS=15 bash scripts/gpu_prof_pmc.sh 2>&1 | tail -7 | tee -a $O/summary.txt
fi
