#!/bin/bash
# Round-3 final payload (LAST GPU action of the round, on the final commit): GPU suite + smoke, the bench line of every
# BASELINE configuration, then rocprofv3 trace + PMC passes (tuned table loaded, --no-autotune: no tuner launches in the
# averages) for config 2 fp32, config 2 x3 and config 3 fp32, and the PMC traffic table bench.py replays.
mkdir -p gpurun_out
export PYTHONDONTWRITEBYTECODE=1
O=gpurun_out
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/final_pytest.log 2>&1; echo "pytest exit $?"; tail -2 $O/final_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1; echo "smoke exit $?"; tail -1 $O/final_smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_bench_cfg2.json 2> $O/final_bench_cfg2.err; echo "bench exit $?"; tail -1 $O/final_bench_cfg2.json | cut -c1-300
for w in cfg1 cfg3 cfg4 cfg5; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 5 > $O/final_bench_$w.json 2> $O/final_bench_$w.err; echo "$w exit $?"; tail -1 $O/final_bench_$w.json | cut -c1-220
done
W=cfg2 TAG=_final STEPS=15 bash scripts/gpu_prof_pmc.sh > $O/final_prof_cfg2.log 2>&1; tail -2 $O/final_prof_cfg2.log
W=cfg2 TAG=_final_x3 STEPS=15 ENVS="PTX_PRECISION=x3" PASSES="1 3 4" bash scripts/gpu_prof_pmc.sh > $O/final_prof_cfg2_x3.log 2>&1; tail -2 $O/final_prof_cfg2_x3.log
W=cfg3 TAG=_final STEPS=15 PASSES="1 3 4" bash scripts/gpu_prof_pmc.sh > $O/final_prof_cfg3.log 2>&1; tail -2 $O/final_prof_cfg3.log
python scripts/pmc_traffic_json.py $O/prof_cfg2_final/summary.txt $O/final_pmc_traffic.json "$PTX_COMMIT" "python bench.py --workload cfg2 --steps 15 --warmup 2 --no-cpu-baseline --no-x3 --no-autotune"
cp $O/final_pmc_traffic.json profiles/r03_pmc_traffic.json   # (on the box: the after-profile line below cites THIS commit's counters)
# the bench line of the SAME build right after its profile (the stem row's average must agree with roofline.avg_launch_ms)
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-x3 --no-cpu-baseline --no-autotune > $O/final_bench_cfg2_after_prof.json 2>/dev/null; tail -1 $O/final_bench_cfg2_after_prof.json | cut -c1-200
