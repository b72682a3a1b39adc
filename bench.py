#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): resnet3d50 (Moments-339) forward, 8x3x16x224x224 synthetic
clips per GPU, clips/sec + max|dlogits| vs the CPU path.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU; a step = one forward of this rank's 8 clips (inputs resident in HBM) plus,
for N > 1, the single all-gather of logits (RCCL).  Weak scaling: 8 clips per GPU.
Rank 0 prints ONE JSON line.

`--workload` selects another BASELINE.json configuration through the same harness (same JSON schema;
the default, and what the driver runs, is config 2):
    cfg1  resnet18 2-D, 1x3x224x224 (plumbing case)      cfg3  (2+1)D-50 + NL blocks, 8x3x32x112x112
    cfg4  I3D, 2x3x64x224x224 per GPU (16 clips / 8 GPUs)  cfg5  BigGAN-deep-256 generator, batch 64, fp16 MFMA operands
    cfg5-fp32  the same generator on the fp32 path
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# what the workloads are and what their rooflines are priced against lives in bench_workloads.py; the per-kernel roofline
# bookkeeping and the N = 1 secondary legs in bench_roofline.py.  This file: the CLI, the bench contract's timed region
# (timed_steps) and the assembly of the JSON line.
from bench_workloads import (CLASSES, CLIPS_PER_GPU, FRAMES, GFLOP_PER_CLIP, GLOBAL_BATCH, PEAK_F16_MFMA_TF,      # noqa: E402,F401
                             PEAK_F32_MFMA_TF, PEAK_HBM_GBS, SIZE, SUSTAINED_F16_MFMA_TF, headline_workload, other_workload,
                             standin_workload)
from bench_roofline import newest_traffic_file as _newest_traffic_file      # noqa: E402,F401  (tests look it up here)


def local_batch(make, per_gpu, workload, scaling, world, rank):
    """This rank's input batch and the job's total units per step.
    weak   (default; the headline's "8 clips per GPU"): every rank draws its own `per_gpu` units (seed 99 + rank);
           total = per_gpu x N.
    strong (SURVEY.md 8e "report it too if cheap"; config 4 is strong by construction, 16 clips over 8 GPUs): ONE global
           batch of BASELINE.json's size, identical on every rank (seed 99), cut into contiguous dim-0 shards exactly as the
           reference's DataParallel scatter does (parallel.shard_clips; ragged when N does not divide it).  At N = 8 the
           headline's 8-clip batch leaves ONE clip per GPU.
    `make(n, seed)` builds n units on the CPU."""
    from pretorched_x_amd.parallel import shard_clips
    if scaling == "strong":
        if workload not in GLOBAL_BATCH:
            raise SystemExit("--scaling strong: %s has no clip batch to shard" % workload)
        total = GLOBAL_BATCH[workload]
        return shard_clips(make(total, 99), world, rank), total
    return make(per_gpu, 99 + rank), per_gpu * world


def timed_steps(run, total_units, steps, warmup, dev, sync):
    """The timed region of the bench contract -- W untimed warm-up steps, then exactly K steps bracketed by a
    barrier + device synchronisation on both sides, MAX over ranks -- followed, for N > 1, by the self-check of
    the clip-parallel step (parallel.verify_gather) OUTSIDE the timed region.  Backend-agnostic: the CPU test
    drives this very function under gloo with a stand-in forward.
    A step = one forward of this rank's units (+ the one all-gather of logits when N > 1).  `total_units`: units of
    the WHOLE job per step (weak scaling: units per GPU x N; strong scaling: the fixed global batch, shards may be ragged).
    Returns (elapsed seconds [MAX over ranks], last step output, verify dict or None, per-rank timing dict): the MAX hides
    WHICH rank was slow, so every rank's own time is all-gathered too -- {"ms_per_step": [rank 0 .. N-1], "min", "max",
    "argmax_rank", "spread"} -- and a straggler in an 8-GPU run is visible in the line itself."""
    import torch.distributed as dist
    from pretorched_x_amd.parallel import gather_logits, verify_gather
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1

    def step():
        out = run()
        if world > 1 and out.dim() == 2:           # class logits: the path's one collective (images stay sharded)
            out = gather_logits(out, total=total_units)
        return out

    out = None
    for _ in range(warmup):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    verify = None
    per_rank = [elapsed]
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        per_rank = [float(e.item()) for e in every]
        elapsed = max(per_rank)
        local = run()
        if local.dim() == 2:
            verify = verify_gather(local, gather_logits(local, total=total_units))
            # a second, independent forward must reproduce the timed steps' result bit for bit
            verify["deterministic"] = bool(torch.equal(gather_logits(local, total=total_units), out))
            flag = torch.tensor([int(verify["deterministic"])], device=dev, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            verify["deterministic"] = bool(flag.item())
    ms = [1e3 * e / max(steps, 1) for e in per_rank]
    rank_ms = {"ms_per_step": [round(v, 4) for v in ms], "min": round(min(ms), 4), "max": round(max(ms), 4),
               "argmax_rank": int(max(range(len(ms)), key=lambda i: ms[i])),
               "spread": round((max(ms) - min(ms)) / max(max(ms), 1e-12), 4)}
    return elapsed, out, verify, rank_ms


def self_launch(n):
    """Replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same args>`.
    Fails loudly when the node has fewer than n GPUs (PTX_BENCH_BACKEND=gloo -- the functional check in which ranks share
    devices -- is exempt)."""
    import torch
    have = torch.cuda.device_count()
    if os.environ.get("PTX_BENCH_BACKEND", "nccl") == "nccl" and have < n:
        raise SystemExit("bench.py --gpus %d: this node has %d visible GPU(s); one rank per GPU is required "
                         "(PTX_BENCH_BACKEND=gloo runs a functional check with ranks sharing devices)" % (n, have))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    # --standalone: the agent binds its own free rendezvous port (c10d store on port 0) -- nothing is picked here and
    # re-bound later; 127.0.0.1 because the container's hostname may not resolve
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    sys.stderr.flush()
    os.execv(sys.executable, cmd)


def measure(args, scaling, world, rank, local, dev, backend, first=True):
    """One bench line (a dict on rank 0, None elsewhere) for `scaling` in {"weak", "strong"}."""
    import torch.distributed as dist
    standin = os.environ.get("PTX_BENCH_STANDIN") == "1"
    headline = args.workload == "cfg2"
    f16 = args.workload == "cfg5"
    tolerance = 5e-2 if f16 else 1e-3         # fp16 operands: builder-chosen bound on |d image| (parity unpinned)
    if standin:
        model, recipe, make, per_gpu, fwd, cpu_fn, unit, workload_label, sample_idx = standin_workload(args.workload)
    elif headline:
        model, recipe, make, per_gpu, fwd, cpu_fn, unit, workload_label, sample_idx = headline_workload()
    else:
        model, recipe, make, per_gpu, fwd, cpu_fn, unit, workload_label, sample_idx = other_workload(args.workload, rank)
    x_cpu, total_units = local_batch(make, per_gpu, args.workload, scaling, world, rank)
    units_per_gpu = x_cpu.shape[0]                # this rank's share (strong scaling: may be ragged, may be 0 past the batch)
    if units_per_gpu == 0 and not standin:
        raise SystemExit("--scaling strong: %d ranks for a %d-%s batch leaves rank %d without work" % (world, total_units, unit, rank))
    if scaling == "strong":
        workload_label += " -- STRONG scaling: one %d-%s global batch sharded over %d GPU(s), %d on rank 0" % (
            total_units, unit[:-1], world, units_per_gpu)

    sd, eng, tuned_entries = None, None, None
    if standin:
        # functional check of the multi-rank plumbing WITHOUT a GPU: the same local_batch / timed_steps / gather / per-rank
        # bookkeeping as a real run, a CPU stand-in forward, no tuner, no roofline (tests/test_parallel_gloo.py)
        x = x_cpu
        run = lambda: model(x)       # noqa: E731
        sync = lambda: None          # noqa: E731
    else:
        import pretorched_x_amd as ptx
        from pretorched_x_amd.parallel import broadcast_tuned_table
        from pretorched_x_amd.testing import synth_state_dict
        sd = synth_state_dict(model.state_dict(), 1234, **recipe)
        model.load_state_dict(sd)
        model = model.to(dev).eval()
        model.engine().check_weights = True
        x = x_cpu.to(dev)
        run = (lambda: model(x)) if fwd is None else (lambda: fwd(model, x))
        sync = torch.cuda.synchronize
        eng = model.engine()
        if args.lanes is not None:
            eng.lanes = args.lanes
        full_tune = os.environ.get("PTX_FULL_TUNE") == "1" and fwd is None
        if not args.no_autotune and first:
            # tile configurations are timed on rank 0 ONLY and broadcast: N tuners running at once on one node
            # perturb each other's HIP-event timings, and every rank must launch the same kernels
            if rank == 0:
                if headline or full_tune:
                    eng.autotune(model, x, iters=int(os.environ.get("PTX_TUNE_ITERS", "2")), verbose=args.verbose)      # every candidate tile of every conv problem
                else:
                    run()                              # first call compiles the plan and times untuned tiles
                    if fwd is None and eng.lanes == "auto":
                        # ... and what the table lacks of this workload: the clip-lanes verdict of its (architecture, shape)
                        eng.autotune(model, x, iters=int(os.environ.get("PTX_TUNE_ITERS", "2")), verbose=args.verbose, only_untuned=True)
                torch.cuda.synchronize()
            if world > 1:
                # the other ranks sit in this broadcast while rank 0 tunes: the process group's timeout (PTX_BENCH_TIMEOUT,
                # default 1800 s) bounds that wait; say WHAT was being waited for instead of a bare collective timeout
                try:
                    tuned_entries = broadcast_tuned_table(src=0)
                except Exception as e:     # noqa: BLE001
                    raise SystemExit("bench.py rank %d: the broadcast of rank 0's tuned tile table failed or timed out after "
                                     "PTX_BENCH_TIMEOUT=%ss -- rank 0's autotune outlasted the other ranks' wait (raise the "
                                     "timeout or pass --no-autotune): %s" % (rank, os.environ.get("PTX_BENCH_TIMEOUT", "1800"), e))
                eng.invalidate()                       # plans pick their tiles at compile time: recompile with the table
            if headline and rank == 0 and os.environ.get("PTX_TUNED_OUT"):
                from pretorched_x_amd.engine import save_tuned_table
                save_tuned_table(os.environ["PTX_TUNED_OUT"])

    elapsed, out, verify, rank_ms = timed_steps(run, total_units, args.steps, args.warmup, dev, sync)
    n_dev = torch.cuda.device_count()
    me = {"rank": rank, "local_rank": local, "device": torch.cuda.current_device() if n_dev and not standin else None,
          "name": torch.cuda.get_device_name(local) if n_dev and not standin else "cpu", "pid": os.getpid(),
          "units": units_per_gpu, "plan_builds": eng.plan_builds if eng is not None else None}
    ranks_seen = {"world_size": 1, "device_count": n_dev, "distinct_devices": 1, "ranks": [me],
                  "tuned_entries_broadcast": None}
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, me)
        ranks_seen = {"world_size": dist.get_world_size(), "device_count": n_dev,
                      "distinct_devices": len({r["device"] for r in seen}), "ranks": seen,
                      "tuned_entries_broadcast": tuned_entries}
        if backend == "nccl" and ranks_seen["distinct_devices"] != world:
            raise SystemExit("bench.py: %d ranks on %d distinct GPUs -- one rank per GPU is the contract" % (
                world, ranks_seen["distinct_devices"]))

    ms_per_step = 1e3 * elapsed / args.steps
    clips_per_s = total_units * args.steps / elapsed
    if rank != 0:
        return None

    parallelism = "clip-parallel x%d, one all-gather of logits" % world
    if standin:
        parallelism += " -- FUNCTIONAL CHECK: CPU stand-in forward over %s, nothing here is a measurement" % backend
    elif backend != "nccl" and world > 1:
        parallelism += " -- FUNCTIONAL CHECK over %s, ranks sharing %d device(s): not a scaling figure" % (backend, n_dev)
    result = {
        "metric": ("clips/sec, resnet3d50 forward 8x3x16x224x224 per GPU (+ max|dlogits| vs CPU)" if headline else
                   "%s/sec, %s (+ max|d output| vs CPU)" % (unit, args.workload)),
        "value": round(clips_per_s, 2), "unit": "%s/s" % unit, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": scaling, "vs_baseline": None, "dtype": "f16" if f16 else "f32", "data": "synthetic",
        "config": {"workload": workload_label, "clips_per_gpu": units_per_gpu, "global_batch": total_units,
                   "parallelism": parallelism},
    }
    if standin:
        result.update({"standin": True, "roofline": None, "cpu_baseline": None, "distributed_check": verify,
                       "ranks_seen": ranks_seen, "rank_ms_per_step": rank_ms})
        return result

    import bench_roofline as R
    import pretorched_x_amd as ptx
    lanes_used = eng.lanes_for(units_per_gpu, model, x.shape) if fwd is None else 1
    result["config"]["clip_lanes"] = lanes_used

    # ---- CPU baseline: the oracle restatement of the reference path on this box's host cores ----
    def sample_indices():
        """Units of this rank's batch that the CPU oracle recomputes: the full batch for config 2; else a bounded
        sample -- the workload's own picks (config 5: images from BOTH 32-image chunks Engine.generate runs) or the
        first two."""
        if headline:
            return list(range(units_per_gpu))
        if sample_idx is not None:
            return [i for i in sample_idx if i < units_per_gpu]
        return list(range(min(2, units_per_gpu)))
    cpu, parity, want, idx = None, None, None, sample_indices()
    if not args.no_cpu_baseline and world > 1:
        parity, want = R.shard_parity(run, cpu_fn, sd, x_cpu, idx, tolerance, world, unit)
    if not args.no_cpu_baseline and world == 1 and first:      # reported at N = 1 only (bench contract)
        cpu, parity, want = R.cpu_baseline_leg(run, cpu_fn, sd, x_cpu, idx, tolerance, headline, units_per_gpu, unit)

    # ---- the other clip-lane setting: the headline ran `lanes_used` lanes (Engine.lanes, default "auto" = what the tuned
    # table holds for this model and shape); the leg runs the other one of {1, 2}, so both figures are in every N = 1 line
    lanes_leg = None
    if world == 1 and first and fwd is None and units_per_gpu % 2 == 0 and (lanes_used > 1 or not args.no_lanes):
        other = 1 if lanes_used > 1 else 2
        lanes_leg = R.lanes_leg(args, eng, model, x, run, headline, full_tune, units_per_gpu, unit, clips_per_s / world,
                                "headline_%d_lane%s" % (lanes_used, "" if lanes_used == 1 else "s"), out, want, idx, tolerance,
                                None, other)

    # ---- per-kernel roofline: every launch of the SINGLE-PLAN path timed with HIP events on the launch stream.  When the
    # headline ran as clip lanes, its launches overlap on two streams and have no individual durations to compare with a
    # rocprofv3 trace; the per-kernel section then describes the one-plan execution of the same batch (the lanes leg above
    # timed it: `launch_timing.ms_per_step`), `roofline_net` stays the headline's whole-job rate
    keep_lanes = eng.lanes
    eng.lanes = 1
    single_ms = ms_per_step
    plan = None
    if lanes_used > 1:
        single_ms = lanes_leg["ms_per_step"] if lanes_leg and lanes_leg["lanes"] == 1 else None
    if lanes_leg is not None:
        with torch.cuda.device(dev):
            plan = eng.plan_for(model, x)          # the full-batch plan (the lanes leg left half-batch plans behind it)
    fields, gflop_per_unit = R.kernel_rooflines(eng, model, dev, args.workload, f16, GFLOP_PER_CLIP if headline else None,
                                                clips_per_s / world, ms_per_step, plan=plan, single_plan_ms=single_ms)
    if lanes_used > 1:
        fields["launch_timing"]["note"] = ("the headline ran %d clip lanes; these per-launch figures are the single-plan execution "
                                           "of the same batch, whose step time is launch_timing.ms_per_step" % lanes_used)
    result.update(fields)

    split = None
    if world == 1 and first and not f16 and args.workload != "cfg5-fp32" and not args.no_x3:
        # the split-operand leg is a single-plan figure
        split = R.x3_leg(args, eng, model, x, run, dev, headline, full_tune, units_per_gpu, unit, gflop_per_unit,
                         clips_per_s / world, want, idx, tolerance)
    eng.lanes = keep_lanes
    result.update({
        "cpu_baseline": cpu, "parity": parity, "split_f16x3": split, "clip_lanes": lanes_leg,
        "commit": os.environ.get("PTX_COMMIT"),
        # which sources the loaded libptx_amd.so was compiled from, and whether that is this tree (build.py stamps it)
        "binary": {"version": ptx._lib.lib().ptx_version().decode(), "source_sha256_matches_tree": ptx._lib.binary_source_hash() == ptx._lib.source_hash()},
        "distributed_check": verify, "ranks_seen": ranks_seen, "rank_ms_per_step": rank_ms,
    })
    return result


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks (one per GPU).  Default: the launcher's WORLD_SIZE when there is one, else 1; an explicit value "
                         "that disagrees with the launcher is an error")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--no-x3", action="store_true", help="skip the secondary split-precision (x3) leg")
    ap.add_argument("--no-lanes", action="store_true", help="skip the secondary clip-lanes leg (the lane setting the headline did not run)")
    ap.add_argument("--lanes", type=int, default=None, help="force Engine.lanes for the headline (default: the engine's own, 'auto' = tuned table)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg1", "cfg2", "cfg3", "cfg4", "cfg5", "cfg5-fp32"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong", "both"],
                    help="weak (default, the headline): a fixed batch PER GPU; strong: BASELINE's global batch (cfg2 / cfg3: 8 "
                         "clips, cfg4: 16) sharded over the ranks -- at 8 GPUs the headline batch leaves 1 clip per GPU; "
                         "both: the weak line, then the strong line, in one invocation")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: become N ranks (one process per GPU, RCCL) by re-executing under
        # torch.distributed.run -- the same command line the driver uses.  The N = 1 path never gets here.
        self_launch(args.gpus)

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # PTX_BENCH_BACKEND=gloo: a functional check of the N > 1 branches on a box with FEWER GPUs than ranks (ranks share
    # devices round-robin, collectives go through gloo) -- never a scaling figure; the line says so in `config.parallelism`
    backend = os.environ.get("PTX_BENCH_BACKEND", "nccl")
    import datetime
    pg_timeout = datetime.timedelta(seconds=int(os.environ.get("PTX_BENCH_TIMEOUT", "1800")))
    scalings = ["weak", "strong"] if args.scaling == "both" else [args.scaling]
    if os.environ.get("PTX_BENCH_LAUNCH_CHECK") == "1":
        # launcher check (runs without GPUs, tests/test_parallel_gloo.py): the ranks `--gpus N` started rendezvous over gloo,
        # report who they are, rank 0 prints the launch-related fields of the line(s) a real run would print -- one per
        # scaling, in the order they would be measured -- nothing is measured
        if args.gpus != world:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d rank(s); they must agree" % (args.gpus, world))
        if world > 1:
            dist.init_process_group("gloo", timeout=pg_timeout)
        seen = [None] * world
        me = {"rank": rank, "local_rank": local, "pid": os.getpid()}
        if world > 1:
            dist.all_gather_object(seen, me)
            dist.barrier()
        else:
            seen = [me]
        if rank == 0:
            for sc in scalings:
                print(json.dumps({"launch_check": True, "n_gpus": world, "scaling": sc, "process_group_timeout_s": pg_timeout.total_seconds(),
                                  "ranks_seen": {"world_size": dist.get_world_size() if world > 1 else 1,
                                                 "distinct_pids": len({r["pid"] for r in seen}),
                                                 "local_ranks": sorted(r["local_rank"] for r in seen)}}))
        if world > 1:
            dist.destroy_process_group()
        return
    # PTX_BENCH_STANDIN=1 (with PTX_BENCH_BACKEND=gloo): the same N-rank code path -- local_batch, timed_steps, the gather,
    # the per-rank bookkeeping, one line per scaling mode -- with a CPU stand-in forward and no GPU at all: how the 8-rank
    # regimes of the metric are exercised where there are no 8 GPUs (tests/test_parallel_gloo.py).  Never a measurement.
    standin = os.environ.get("PTX_BENCH_STANDIN") == "1"
    if standin and backend == "nccl":
        raise SystemExit("bench.py: PTX_BENCH_STANDIN=1 is the CPU functional check; it needs PTX_BENCH_BACKEND=gloo")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if not standin:
            torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d rank(s); they must agree" % (args.gpus, world))
    if standin:
        dev = torch.device("cpu")
    else:
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)

    results = []
    for i, scaling in enumerate(scalings):
        # --scaling both: the weak line (the headline), then the strong one, from the same ranks in one invocation; the
        # second pass reuses the first's tuned tiles and skips the N = 1 extras (CPU baseline timing, split-operand leg)
        results.append(measure(args, scaling, world, rank, local, dev, backend, first=(i == 0)))
    if rank == 0 and os.environ.get("PTX_TUNED_OUT") and not standin:      # tile choices of this run (both legs), for tuned_gfx950.json
        from pretorched_x_amd.engine import save_tuned_table
        save_tuned_table(os.environ["PTX_TUNED_OUT"])
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        for result in results:
            print(json.dumps(result))


if __name__ == "__main__":
    main()
