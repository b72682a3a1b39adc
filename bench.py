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

CLIPS_PER_GPU, FRAMES, SIZE, CLASSES = 8, 16, 224, 339
GFLOP_PER_CLIP = 79.692           # 2 x 318.768 GMAC / 8 clips, padding taps counted
PEAK_F32_MFMA_TF = 157.3          # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TF = 2500.0         # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
SUSTAINED_F16_MFMA_TF = 1600.0     # measured: scripts/micro/mfma_f16_peak.hip, random operands (profiles/r03_mfma_f16_peak.txt)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md HBM3E peak
def _newest_traffic_file(workload="cfg2"):
    """Newest committed PMC traffic table of `workload`, replayed in roofline.traffic: profiles/rNN_pmc_traffic_<workload>.json,
    or -- config 2 only, the table's original name -- profiles/rNN_pmc_traffic.json.  None when the workload has no table:
    another workload's launches of the same tile are a different problem, their counters are never replayed."""
    import glob
    pats = ["r[0-9][0-9]_pmc_traffic_%s.json" % workload] + (["r[0-9][0-9]_pmc_traffic.json"] if workload == "cfg2" else [])
    names = sorted(os.path.basename(f) for pat in pats for f in glob.glob(os.path.join(ROOT, "profiles", pat)))
    return names[-1] if names else None


def other_workload(name, rank):
    """(model, weight recipe, make(n, seed) -> n CPU units, units per GPU, forward(model, device input) or None,
    cpu oracle fn(sd, units, idx), unit, label, parity sample indices or None) for the non-headline BASELINE.json
    configurations.  `idx`: positions of the sampled units inside this rank's batch (labels of config 5 follow them)."""
    import pretorched_x_amd as ptx
    from oracle import functional as OF
    from pretorched_x_amd.testing import BIGGAN_RECIPE, I3D_RECIPE, synth_state_dict

    def randn(*shape):
        return lambda n, seed: torch.randn(n, *shape, generator=torch.Generator().manual_seed(seed))
    if name == "cfg1":
        m = ptx.resnet18(num_classes=1000, pretrained=None)
        return m, {}, randn(3, 224, 224), 1, None, lambda sd, x, idx: OF.forward(OF.ARCHS["resnet18"], sd, x), "images", \
            "resnet18 2-D forward, 1x3x224x224 (config 1; arithmetic reference = torchvision stand-in, parity unpinned)", None
    if name == "cfg3":
        m, recipe = ptx.nonlocal_r2plus1d50(339), dict(inner_bn_damp=0.9, nl_bn_damp=0.05)
        return m, recipe, randn(3, 32, 112, 112), 8, None, \
            lambda sd, x, idx: OF.forward(OF.ARCHS["nonlocal_r2plus1d50"], sd, x), "clips", \
            "resnet2p1d50 + NL blocks forward, 8x3x32x112x112 synthetic clips per GPU (config 3)", None
    if name == "cfg4":
        from oracle import i3d_standin as I3
        m, recipe = ptx.i3d(400), I3D_RECIPE
        return m, recipe, randn(3, 64, 224, 224), 2, None, lambda sd, x, idx: I3.forward(sd, x), "clips", \
            "I3D (InceptionV1-3D) forward, 2x3x64x224x224 synthetic clips per GPU = 16 over 8 GPUs (config 4; parity unpinned)", None
    if name in ("cfg5", "cfg5-fp32"):
        from oracle import biggan_standin as BG
        half = name == "cfg5"
        m, recipe = ptx.biggan_deep(256, precision="fp16" if half else "fp32"), BIGGAN_RECIPE
        g = torch.Generator().manual_seed(99 + rank)
        z = torch.randn(64, 128, generator=g)
        lab = torch.randint(0, 1000, (64,), generator=g)

        def fwd(model, zd, lab=lab):
            return model(zd, model.shared(lab.to(zd.device)))
        # Engine.generate runs batch 64 as two 32-image chunks: the parity sample takes images from BOTH
        return m, recipe, (lambda n, seed: z[:n]), 64, fwd, \
            lambda sd, zs, idx: BG.forward(sd, zs, sd["shared.weight"][lab[idx]]), "images", \
            ("BigGAN-deep-256 generator, batch 64 z ~ N(0,1) + class labels per GPU, %s (config 5; parity unpinned)" %
             ("fp16 MFMA operands, fp32 accumulate / skip / output" if half else "fp32 MFMA path")), [0, 31, 32, 63]
    raise SystemExit("unknown workload %r" % name)


GLOBAL_BATCH = {"cfg1": 1, "cfg2": 8, "cfg3": 8, "cfg4": 16}      # BASELINE.json configs: the batch the metric is quoted on


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
    import pretorched_x_amd as ptx
    from pretorched_x_amd.parallel import broadcast_tuned_table
    from pretorched_x_amd.testing import synth_clips, synth_state_dict

    headline = args.workload == "cfg2"
    f16 = args.workload == "cfg5"
    peak_tf = PEAK_F16_MFMA_TF if f16 else PEAK_F32_MFMA_TF
    tolerance = 5e-2 if f16 else 1e-3         # fp16 operands: builder-chosen bound on |d image| (parity unpinned)
    fwd, unit, sample_idx = None, "clips", None
    if headline:
        model = ptx.__dict__["resnet3d50"](num_classes=CLASSES, pretrained=None)
        sd = synth_state_dict(model.state_dict(), 1234)
        make, per_gpu = (lambda n, seed: synth_clips(n, FRAMES, SIZE, seed)), CLIPS_PER_GPU      # per-rank clips
        from oracle import functional as OF_
        cpu_fn = lambda sd_, x_, idx_: OF_.forward(OF_.ARCHS["resnet3d50"], sd_, x_)   # noqa: E731
        workload_label = ("resnet3d50 (Moments-339) forward, %dx3x%dx%dx%d synthetic clips per GPU, "
                          "random-init weights (seeded recipe)" % (CLIPS_PER_GPU, FRAMES, SIZE, SIZE))
    else:
        model, recipe, make, per_gpu, fwd, cpu_fn, unit, workload_label, sample_idx = other_workload(args.workload, rank)
        sd = synth_state_dict(model.state_dict(), 1234, **recipe)
    x_cpu, total_units = local_batch(make, per_gpu, args.workload, scaling, world, rank)
    units_per_gpu = x_cpu.shape[0]                # this rank's share (strong scaling: may be ragged, may be 0 past the batch)
    if units_per_gpu == 0:
        raise SystemExit("--scaling strong: %d ranks for a %d-%s batch leaves rank %d without work" % (world, total_units, unit, rank))
    if scaling == "strong":
        workload_label += " -- STRONG scaling: one %d-%s global batch sharded over %d GPU(s), %d on rank 0" % (
            total_units, unit[:-1], world, units_per_gpu)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    model.engine().check_weights = True
    x = x_cpu.to(dev)
    run = (lambda: model(x)) if fwd is None else (lambda: fwd(model, x))

    eng = model.engine()
    tuned_entries = None
    if not args.no_autotune and first:
        # tile configurations are timed on rank 0 ONLY and broadcast: N tuners running at once on one node
        # perturb each other's HIP-event timings, and every rank must launch the same kernels
        if rank == 0:
            if headline or (os.environ.get("PTX_FULL_TUNE") == "1" and fwd is None):
                eng.autotune(model, x, iters=int(os.environ.get("PTX_TUNE_ITERS", "2")), verbose=args.verbose)      # every candidate tile of every conv problem
            else:
                run()                              # first call compiles the plan and times untuned tiles
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

    elapsed, out, verify, rank_ms = timed_steps(run, total_units, args.steps, args.warmup, dev, torch.cuda.synchronize)
    me = {"rank": rank, "local_rank": local, "device": torch.cuda.current_device(),
          "name": torch.cuda.get_device_name(local), "pid": os.getpid(), "plan_builds": eng.plan_builds}
    ranks_seen = {"world_size": 1, "device_count": torch.cuda.device_count(), "distinct_devices": 1, "ranks": [me],
                  "tuned_entries_broadcast": None}
    if world > 1:
        seen = [None] * world
        dist.all_gather_object(seen, me)
        ranks_seen = {"world_size": dist.get_world_size(), "device_count": torch.cuda.device_count(),
                      "distinct_devices": len({r["device"] for r in seen}), "ranks": seen,
                      "tuned_entries_broadcast": tuned_entries}
        if backend == "nccl" and ranks_seen["distinct_devices"] != world:
            raise SystemExit("bench.py: %d ranks on %d distinct GPUs -- one rank per GPU is the contract" % (
                world, ranks_seen["distinct_devices"]))

    ms_per_step = 1e3 * elapsed / args.steps
    clips_per_s = total_units * args.steps / elapsed

    result = None
    if rank == 0:
        # ---- per-kernel roofline: every conv launch timed with HIP events on the launch stream ----
        plan = list(eng._plans.values())[-1]
        plan.bind(model)
        with torch.cuda.device(dev):
            all_rows = eng.profile_steps(plan, iters=5)         # EVERY launch of the plan, convs and HBM passes alike
        conv_rows = [r for r in all_rows if r[1] == "conv"]          # (label, kind, bytes, macs, ms, tile, ConvStep)
        all_rows = [r[:6] for r in all_rows]
        rows = [(r[0], r[3], r[4], r[5], r[6].split) for r in conv_rows]
        if os.environ.get("PTX_BENCH_ROWS"):       # per-launch detail for tuning sessions
            with open(os.environ["PTX_BENCH_ROWS"], "w") as f:
                for (label, macs, ms, cfg, split), stp in zip(rows, [r[6] for r in conv_rows]):
                    f.write("%-34s M=%-9d N=%-5d K=%-6d %-28s split=%d %8.4f ms %8.1f TF\n" % (
                        label, stp.d.N * stp.d.To * stp.d.Ho * stp.d.Wo, stp.d.Co,
                        stp.d.Kc * stp.d.kT * stp.d.kH * stp.d.kW, cfg, split, ms, 2e-9 * macs / ms))
                for lab, kind, nb, macs, ms, _cfg in all_rows:
                    if kind == "chain":
                        f.write("%-34s chain %-40s %8.4f ms %8.1f TF\n" % (lab, _cfg, ms, 2e-9 * macs / ms))
                    elif kind != "conv":
                        f.write("%-34s %-5s bytes=%-12d macs=%-14d %8.4f ms %8.1f GB/s\n" % (lab, kind, nb, macs, ms, nb / ms / 1e6))
        # the direct stem (ptx_conv_stem_f32_fwd) is a conv too, with its own kernel
        # ... and so are the chained launches (two convs in one kernel, conv_chain.hip)
        stem_rows = [(lab, macs, ms, cfg, 1) for lab, kind, nb, macs, ms, cfg in all_rows if kind in ("stem", "chain")]
        by_kernel = {}
        for label, macs, ms, cfg, split in rows + stem_rows:
            k = by_kernel.setdefault(cfg, dict(ms=0.0, flop=0.0, launches=0))
            k["ms"] += ms
            k["flop"] += 2.0 * macs
            k["launches"] += 1
        dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
        achieved = dom["flop"] / (dom["ms"] * 1e-3) / 1e12
        conv_ms = sum(v["ms"] for v in by_kernel.values())
        # HBM-bound passes (fold, max-pool, cBN / affine passes ...): algorithmic bytes (compulsory reads + writes of one
        # launch) / HIP-event time of that launch, against the 8 TB/s HBM3E peak
        roofline_hbm, other_ms = {}, 0.0
        for lab, kind, nb, macs, ms, _ in all_rows:
            if kind == "mem":
                h = roofline_hbm.setdefault(lab, dict(ms=0.0, bytes=0, launches=0))
                h["ms"] += ms
                h["bytes"] += nb
                h["launches"] += 1
            elif kind not in ("conv", "stem", "chain"):
                other_ms += ms
        roofline_hbm = {k: {"bound": "hbm", "launches": v["launches"], "ms": round(v["ms"], 4),
                            "algorithmic_MB": round(v["bytes"] / 1e6, 2),
                            "achieved": round(v["bytes"] / v["ms"] / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(v["bytes"] / v["ms"] / 1e6 / PEAK_HBM_GBS, 4)} for k, v in roofline_hbm.items()}
        # HBM-side traffic of the dominant kernel: NOT measured by this command (PMC passes need rocprofv3 around
        # it).  The value replayed here comes from the committed PMC summary named in `traffic_source`, collected in
        # separate --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command; KiB per dispatch, UNCORRECTED
        # (MI355X_MICROARCH.md: FETCH_SIZE may under-report wide streaming reads by up to 2x on gfx950, so the true
        # figure lies in [traffic, traffic + fetch]); null when that file has no row for the dominant tile.
        def traffic_for(name):
            """(bytes, source) of one kernel from the committed PMC table, or (None, None)."""
            try:
                import re
                tfile = _newest_traffic_file(args.workload)
                if tfile is None:
                    return None, None
                tpath = os.path.join("profiles", tfile)
                tj = json.load(open(os.path.join(ROOT, tpath)))

                meta = tj.get("_meta", {})

                def hit(v):
                    return ((v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024.0,
                            {"file": tpath, "commit": meta.get("commit"), "command": meta.get("command"),
                             "fetch_bytes": v["FETCH_SIZE_KiB"] * 1024.0, "write_bytes": v["WRITE_SIZE_KiB"] * 1024.0,
                             "fetch_correction": "none applied (guide: up to 2x under-report on streaming reads)",
                             "measured_in_this_run": False})
                if name.startswith(("conv_stem", "conv3x3_f16", "conv1x1_skip_f16", "conv1x1_pro_f16")):   # own kernels: one row per name
                    # every template instantiation of that kernel, weighted by its dispatch count: the bench groups them too
                    inst = [v for k, v in tj.items() if k.startswith(name + "_kernel") and isinstance(v, dict)
                            and v.get("WRITE_SIZE_KiB") is not None and v.get("FETCH_SIZE_KiB") is not None]
                    if not inst:
                        return None, None
                    calls = [float(v.get("calls", 1)) for v in inst]
                    mean = {c: sum(v[c] * n for v, n in zip(inst, calls)) / sum(calls) for c in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB")}
                    return hit(mean)
                parts = name.split("/")                   # "64x64x32/2x2/m32/dma[N][/chain][/re]"
                tile, waves, mt = parts[:3]
                rest = parts[3:]
                stage = next((t for t in rest if t.startswith("dma")), "")
                want = [int(v) for v in tile.split("x")] + [int(v) for v in waves.split("x")] + [int(mt[1:])]
                want_dma = stage.startswith("dma")
                want_nstage = int(stage[3:]) if len(stage) > 3 else 2
                want_chain, want_re = "chain" in rest, "re" in rest

                def flag(a):
                    return a in ("t", "true", "1")
                for k, v in tj.items():
                    m = re.match(r"conv_igemm<([^>(]*)", k)         # summarize_prof.py: no blanks, t / f booleans, <= 60 characters
                    if not m or not isinstance(v, dict) or v.get("WRITE_SIZE_KiB") is None or v.get("FETCH_SIZE_KiB") is None:
                        continue
                    targs = [a.strip() for a in m.group(1).split(",")]
                    if len(targs) < 10:
                        continue
                    if (len(targs) > 10 and flag(targs[10])) or (len(targs) > 11 and flag(targs[11])):     # fp32-operand tiles only
                        continue
                    chain = len(targs) > 13 and flag(targs[13])
                    repi = len(targs) > 14 and flag(targs[14])
                    if ([int(a) for a in targs[:6]] == want and flag(targs[8]) == want_dma and int(targs[9]) == want_nstage
                            and chain == want_chain and repi == want_re):
                        return hit(v)
            except Exception:
                pass
            return None, None

        # FLOP the direct stem kernels actually ISSUE per launch (pruned temporal taps excluded, K padded 21 -> 22): the
        # host-side twin of SQ_INSTS_MFMA x 4096 from the committed PMC pass
        issued_by_kernel = {}
        for stp in plan.steps:
            if type(stp).__name__ == "StemF32Step":
                issued_by_kernel[stp.kernel] = issued_by_kernel.get(stp.kernel, 0.0) + stp.issued_flop()

        def roof(name, ms, flop, launches):
            traffic, traffic_source = traffic_for(name)
            tf = flop / (ms * 1e-3) / 1e12
            r = {"bound": "mfma", "kernel": ("%s_kernel" if name.startswith(("conv_stem", "conv3x3_f16", "conv1x1_skip_f16", "conv1x1_pro_f16")) else "conv_igemm_kernel<%s>") % name,
                 "achieved": round(tf, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4),
                 "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": launches,
                 "avg_launch_ms": round(ms / launches, 4), "algorithmic_gflop_per_launch": round(flop / launches / 1e9, 3)}
            # `frac` prices padding taps as work (SURVEY.md 8d allows it); `issued_frac` = MFMA FLOP the kernel really
            # issues / time / peak -- null for the generic tiles, whose tap pruning is decided per tile at run time
            iss = issued_by_kernel.get(name)
            r["issued_gflop_per_launch"] = round(iss / launches / 1e9, 3) if iss else None
            r["issued_frac"] = round(iss / (ms * 1e-3) / 1e12 / peak_tf, 4) if iss else None
            return r

        # `roofline`: the kernel (one template instantiation, as rocprofv3 --stats groups them) with the largest total time
        # per step -- since the stem got under 1.5 ms that can be a tile class with a dozen launches of different problems;
        # `roofline_longest_launch`: the single longest launch of the step (the stem), whose average duration is the
        # one-problem row of the committed rocprofv3 summary
        roofline = roof(dom_name, dom["ms"], dom["flop"], dom["launches"])
        ll = max(rows + stem_rows, key=lambda r: r[2])
        roofline_longest = roof(ll[3], ll[2], 2.0 * ll[1], 1)
        roofline_longest["label"] = ll[0]
        gflop_per_unit = GFLOP_PER_CLIP if headline else sum(2e-9 * r[1] for r in rows + stem_rows) / plan.shape[0]
        net_tf = gflop_per_unit * 1e9 * clips_per_s / world / 1e12
        # `frac` prices padding taps as work (SURVEY.md 8d's convention: layer4's T = 1 3x3x3 convs then "run" above the
        # peak); `issued_frac` is its twin on the FLOP the MFMA instructions of one step really issue (pruned tap planes
        # excluded, tile / K padding included: engine.issued_conv_flop, the stem's issued_flop) over the same step time
        issued_step = sum(t.issued_flop() for t in plan.all_convs() if hasattr(t, "issued_flop"))
        issued_tf = issued_step / (ms_per_step * 1e-3) / 1e12 if f16 is False else None
        roofline_net = {"bound": "mfma", "achieved": round(net_tf, 2), "peak": peak_tf,
                        "unit": "TFLOP/s", "frac": round(net_tf / peak_tf, 4),
                        "issued_gflop_per_step": round(issued_step / 1e9, 3) if issued_tf is not None else None,
                        "issued_frac": round(issued_tf / peak_tf, 4) if issued_tf is not None else None,
                        "conv_ms_sum": round(conv_ms, 3),
                        "per_kernel": {k: {"ms": round(v["ms"], 3), "tflops": round(v["flop"] / v["ms"] / 1e9, 1),
                                           "launches": v["launches"]} for k, v in sorted(by_kernel.items())}}

        # ---- CPU baseline: the oracle restatement of the reference path on this box's host cores ----
        cpu = None
        parity = None

        def sample_indices():
            """Units of this rank's batch that the CPU oracle recomputes: the full batch for config 2; else a bounded
            sample -- the workload's own picks (config 5: images from BOTH 32-image chunks Engine.generate runs) or the
            first two."""
            if headline:
                return list(range(units_per_gpu))
            if sample_idx is not None:
                return [i for i in sample_idx if i < units_per_gpu]
            return list(range(min(2, units_per_gpu)))
        if not args.no_cpu_baseline and world > 1:
            # N > 1: the CPU baseline is reported at N = 1 only (bench contract), but the line stays
            # self-verifying -- rank 0 checks ITS OWN shard against the oracle (one bounded CPU forward)
            idx = sample_indices()
            xs = x_cpu[idx]
            torch.set_num_threads(max(1, min(32, (os.cpu_count() or 1) // max(1, world))))
            want = cpu_fn(sd, xs, idx)
            got = run().cpu()[idx]
            parity = {"max_abs_dlogits": float((got - want).abs().max().item()),
                      "max_abs_logit": float(want.abs().max().item()),
                      "argmax_equal": bool(torch.equal(got.argmax(1), want.argmax(1))) if got.dim() == 2 else None,
                      "tolerance": tolerance, "scope": "rank 0's shard (%d %s) vs the CPU oracle" % (xs.shape[0], unit)}
        if not args.no_cpu_baseline and world == 1 and first:      # reported at N = 1 only (bench contract)
            ncpu = os.cpu_count() or 1
            # bounded sample: the full batch for config 2, at most 2 units for the heavier configurations
            idx = sample_indices()
            xs = x_cpu[idx]
            # pick the thread count that runs the reference path fastest on this host (SMT
            # oversubscription makes oneDNN conv3d collapse), then time it: bounded to ~30 s
            cands = sorted({c for c in (16, 32, 64, 128, ncpu // 2) if 1 <= c <= ncpu})
            best_t, best_n, want = None, None, None
            deadline = time.perf_counter() + 30.0
            for n in cands:
                torch.set_num_threads(n)
                t1 = time.perf_counter()
                want = cpu_fn(sd, xs, idx)
                dt = time.perf_counter() - t1
                if best_t is None or dt < best_t:
                    best_t, best_n = dt, n
                if time.perf_counter() > deadline:
                    break
            torch.set_num_threads(best_n)
            times = [best_t]
            while len(times) < 4 and time.perf_counter() < deadline:
                t1 = time.perf_counter()
                cpu_fn(sd, xs, idx)
                times.append(time.perf_counter() - t1)
            med = sorted(times)[len(times) // 2]
            cpu = {"value": round(xs.shape[0] / med, 3), "unit": "%s/s" % unit, "cores": best_n,
                   "label": "oracle on %d host threads (the fastest of a bounded thread-count sweep; NOT the node's %d "
                            "hardware threads -- oneDNN conv3d collapses under SMT oversubscription)" % (best_n, ncpu),
                   "kind": "port", "sample": "%d timed forwards of %s (median), oracle/ (torch CPU fp32, oneDNN) on %d of "
                   "%d host threads" % (len(times), "the full 8x3x16x224x224 batch" if headline else
                                        "%d of the %d %s of a step" % (xs.shape[0], units_per_gpu, unit), best_n, ncpu)}
            got = run().cpu()[idx]
            parity = {"max_abs_dlogits": float((got - want).abs().max().item()),
                      "max_abs_logit": float(want.abs().max().item()),
                      "argmax_equal": bool(torch.equal(got.argmax(1), want.argmax(1))) if got.dim() == 2 else None,
                      "tolerance": tolerance, "units_checked": [int(i) for i in idx]}

        # ---- secondary leg: the same workload with Engine.precision = "x3" (fp32 operands split into half pairs,
        # three fp16 MFMAs per product block, fp32 accumulate -- fp32-ACCURATE).  Reported next to
        # the headline with its own |d output| vs the CPU path and its own denominator (the fp16 dense MFMA peak / 3
        # issued MFMAs per algorithmic product); the headline `value` above stays the plain fp32-MFMA path.
        split = None
        if world == 1 and first and not f16 and args.workload != "cfg5-fp32" and not args.no_x3:
            eng.precision = "x3"
            if not args.no_autotune:
                if headline or (os.environ.get("PTX_FULL_TUNE") == "1" and fwd is None):
                    eng.autotune(model, x, iters=2, verbose=args.verbose)
                else:
                    run()
            for _ in range(args.warmup):
                out3 = run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out3 = run()
            torch.cuda.synchronize()
            el3 = time.perf_counter() - t0
            plan3 = list(eng._plans.values())[-1]
            plan3.bind(model)
            with torch.cuda.device(dev):
                rows3_full = eng.profile_steps(plan3, iters=5)
                steps3 = [r[6] for r in rows3_full if r[1] == "conv"]
                rows3 = [r[:6] for r in rows3_full]
            if os.environ.get("PTX_BENCH_ROWS"):
                with open(os.environ["PTX_BENCH_ROWS"] + ".x3", "w") as f:
                    for (lab, kind, nb, macs, ms, cfg), stp in zip([r for r in rows3 if r[1] == "conv"], steps3):
                        f.write("%-34s M=%-9d N=%-5d K=%-6d %-28s split=%d %8.4f ms %8.1f TF\n" % (
                            lab, stp.d.N * stp.d.To * stp.d.Ho * stp.d.Wo, stp.d.Co,
                            stp.d.Kc * stp.d.kT * stp.d.kH * stp.d.kW, cfg, stp.split, ms, 2e-9 * macs / ms))
                    for lab, kind, nb, macs, ms, _ in rows3:
                        if kind != "conv":
                            f.write("%-34s %-5s bytes=%-12d macs=%-14d %8.4f ms %8.1f TF\n" % (lab, kind, nb, macs, ms, 2e-9 * macs / ms))
            conv3 = [r for r in rows3 if r[1] == "conv"]
            byk = {}
            # the direct split-operand stem (ptx_conv_stem_x3_fwd) is a conv too, with its own kernel
            for lab, kind, nb, macs, ms, cfg in conv3 + [r for r in rows3 if r[1] == "stem"]:
                k = byk.setdefault(cfg, dict(ms=0.0, flop=0.0, launches=0))
                k["ms"] += ms
                k["flop"] += 2.0 * macs
                k["launches"] += 1
            dn, dv = max(byk.items(), key=lambda kv: kv[1]["ms"])
            rate3 = units_per_gpu * args.steps / el3          # (world == 1 here)
            peak3 = PEAK_F16_MFMA_TF / 3.0
            tf3 = gflop_per_unit * 1e9 * rate3 / 1e12
            split = {"precision": "fp32 operands as half (hi, lo) pairs: a.b = hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_f16, "
                                  "fp32 accumulate; activations / epilogues / outputs fp32",
                     "value": round(rate3, 2), "unit": "%s/s" % unit, "ms_per_step": round(1e3 * el3 / args.steps, 4),
                     "speedup_vs_fp32_mfma": round(rate3 / (clips_per_s / world), 3),
                     "roofline": {"bound": "mfma", "kernel": ("%s_kernel" if dn.startswith(("conv_stem", "conv3x3_f16", "conv1x1_skip_f16", "conv1x1_pro_f16")) else "conv_igemm_kernel<%s>") % dn,
                                  "achieved": round(dv["flop"] / dv["ms"] / 1e9, 2), "peak": round(peak3, 1),
                                  "unit": "TFLOP/s (algorithmic fp32-equivalent; peak = 2500 dense f16 / 3 MFMAs per product)",
                                  "frac": round(dv["flop"] / dv["ms"] / 1e9 / peak3, 4), "launches_per_step": dv["launches"],
                                  "avg_launch_ms": round(dv["ms"] / dv["launches"], 4),
                                  # what a register-only loop of 32x32x16 f16 MFMAs sustains on RANDOM operands (clock / power):
                                  # 1.57-1.65 of the 2.5 PFLOP/s (profiles/r03_mfma_f16_peak.txt); informational, `frac` stays
                                  # against the nominal dense peak
                                  "peak_sustained_random_operands": round(SUSTAINED_F16_MFMA_TF / 3.0, 1),
                                  "frac_of_sustained": round(dv["flop"] / dv["ms"] / 1e9 / (SUSTAINED_F16_MFMA_TF / 3.0), 4)},
                     "roofline_net": {"achieved": round(tf3, 2), "peak": round(peak3, 1), "frac": round(tf3 / peak3, 4),
                                      "vs_fp32_mfma_peak": round(tf3 / PEAK_F32_MFMA_TF, 4),
                                      "conv_ms_sum": round(sum(r[4] for r in rows3 if r[1] in ("conv", "stem", "chain")), 3),
                                      "non_conv_ms": round(sum(r[4] for r in rows3 if r[1] not in ("conv", "stem", "chain")), 3)},
                     "parity": None}
            if parity is not None:
                got3 = out3.cpu()[idx]
                split["parity"] = {"max_abs_dlogits": float((got3 - want).abs().max().item()),
                                   "argmax_equal": bool(torch.equal(got3.argmax(1), want.argmax(1))) if got3.dim() == 2 else None,
                                   "tolerance": tolerance}
            eng.precision = "fp32"

        # ---- secondary leg: the same batch as TWO clip lanes (Engine.lanes = 2: two half-batch plans on two HIP streams, the
        # logits concatenated).  Opt-in in the product, so it is reported NEXT to the headline, which
        # stays the single-plan path every per-kernel figure above describes.
        lanes_leg = None
        if world == 1 and first and not args.no_lanes and units_per_gpu % 2 == 0:
            eng.lanes = 2
            half = x[:units_per_gpu // 2]
            if not args.no_autotune and (headline or os.environ.get("PTX_FULL_TUNE") == "1"):
                eng.autotune(model, half, iters=int(os.environ.get("PTX_TUNE_ITERS", "2")), verbose=args.verbose)
            for _ in range(max(args.warmup, 1)):
                out_l = run()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                out_l = run()
            torch.cuda.synchronize()
            el_l = time.perf_counter() - t0
            rate_l = units_per_gpu * args.steps / el_l
            lanes_leg = {"lanes": 2, "value": round(rate_l, 2), "unit": "%s/s" % unit, "ms_per_step": round(1e3 * el_l / args.steps, 4),
                         "speedup_vs_single_plan": round(rate_l / (clips_per_s / world), 4),
                         "launch_shape": "%d %s per launch (two plans of %d-%s batches, own buffers, tiles tuned for that shape)" % (
                             units_per_gpu // 2, unit, units_per_gpu // 2, unit[:-1]),
                         "max_abs_d_vs_single_plan": float((out_l - out).abs().max().item()),
                         "argmax_equal_single_plan": bool(torch.equal(out_l.argmax(1), out.argmax(1))) if out_l.dim() == 2 else None,
                         "parity": None}
            if parity is not None:
                got_l = out_l.cpu()[idx]
                lanes_leg["parity"] = {"max_abs_dlogits": float((got_l - want).abs().max().item()),
                                       "argmax_equal": bool(torch.equal(got_l.argmax(1), want.argmax(1))) if got_l.dim() == 2 else None,
                                       "tolerance": tolerance}
            if split is not None:
                # ... and the split-operand arithmetic through the same two lanes (its shorter matrix work leaves more of a
                # launch to gaps and HBM-bound passes, so the overlap pays more)
                eng.precision = "x3"
                if not args.no_autotune and (headline or os.environ.get("PTX_FULL_TUNE") == "1"):
                    eng.autotune(model, half, iters=int(os.environ.get("PTX_TUNE_ITERS", "2")), verbose=args.verbose)
                for _ in range(max(args.warmup, 1)):
                    out_l3 = run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    out_l3 = run()
                torch.cuda.synchronize()
                el_l3 = time.perf_counter() - t0
                rate_l3 = units_per_gpu * args.steps / el_l3
                lanes_leg["split_f16x3"] = {"value": round(rate_l3, 2), "unit": "%s/s" % unit, "ms_per_step": round(1e3 * el_l3 / args.steps, 4),
                                            "speedup_vs_single_plan_x3": round(rate_l3 / split["value"], 4), "parity": None}
                if parity is not None:
                    got_l3 = out_l3.cpu()[idx]
                    lanes_leg["split_f16x3"]["parity"] = {
                        "max_abs_dlogits": float((got_l3 - want).abs().max().item()),
                        "argmax_equal": bool(torch.equal(got_l3.argmax(1), want.argmax(1))) if got_l3.dim() == 2 else None,
                        "tolerance": tolerance}
                eng.precision = "fp32"
            eng.lanes = 1

        result = {
            "metric": ("clips/sec, resnet3d50 forward 8x3x16x224x224 per GPU (+ max|dlogits| vs CPU)" if headline else
                       "%s/sec, %s (+ max|d output| vs CPU)" % (unit, args.workload)),
            "value": round(clips_per_s, 2), "unit": "%s/s" % unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f16" if f16 else "f32", "data": "synthetic",
            "config": {"workload": workload_label,
                       "clips_per_gpu": units_per_gpu, "global_batch": total_units,
                       "parallelism": "clip-parallel x%d, one all-gather of logits" % world +
                                      ("" if backend == "nccl" or world == 1 else
                                       " -- FUNCTIONAL CHECK over %s, ranks sharing %d device(s): not a scaling figure" % (
                                           backend, torch.cuda.device_count()))},
            "roofline": roofline, "roofline_longest_launch": roofline_longest, "roofline_net": roofline_net, "roofline_hbm": roofline_hbm,
            "non_conv_ms": round(sum(v["ms"] for v in roofline_hbm.values()) + other_ms, 4),
            "cpu_baseline": cpu, "parity": parity, "split_f16x3": split, "clip_lanes": lanes_leg,
            "commit": os.environ.get("PTX_COMMIT"),
            # which sources the loaded libptx_amd.so was compiled from, and whether that is this tree (build.py stamps it)
            "binary": {"version": ptx._lib.lib().ptx_version().decode(), "source_sha256_matches_tree": ptx._lib.binary_source_hash() == ptx._lib.source_hash()},
            "distributed_check": verify, "ranks_seen": ranks_seen, "rank_ms_per_step": rank_ms,
        }
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
    ap.add_argument("--no-lanes", action="store_true", help="skip the secondary clip-lanes (Engine.lanes = 2) leg")
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
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d rank(s); they must agree" % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    results = []
    for i, scaling in enumerate(scalings):
        # --scaling both: the weak line (the headline), then the strong one, from the same ranks in one invocation; the
        # second pass reuses the first's tuned tiles and skips the N = 1 extras (CPU baseline timing, split-operand leg)
        results.append(measure(args, scaling, world, rank, local, dev, backend, first=(i == 0)))
    if rank == 0 and os.environ.get("PTX_TUNED_OUT"):      # tile choices of this run (both legs), for tuned_gfx950.json
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
