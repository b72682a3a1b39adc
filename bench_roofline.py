"""Roofline bookkeeping and the secondary legs of bench.py (rank 0, N = 1 extras): per-kernel MFMA / HBM rooflines from
HIP-event launch times, the replay of committed rocprofv3 / PMC summaries, the CPU baseline (oracle) leg, the
split-operand ("x3") leg and the clip-lanes leg.  Nothing here is inside the timed region of the bench contract
(bench.timed_steps)."""
import glob
import json
import os
import re
import time

import torch

from bench_workloads import (PEAK_F16_MFMA_TF, PEAK_F32_MFMA_TF, PEAK_HBM_GBS, SUSTAINED_F16_MFMA_TF)

ROOT = os.path.dirname(os.path.abspath(__file__))
OWN_KERNELS = ("conv_stem", "conv3x3_f16", "conv1x1_skip_f16", "conv1x1_pro_f16", "conv_body_f32", "conv_tstack_f32")   # one row per kernel name


def newest_traffic_file(workload="cfg2"):
    """Newest committed PMC traffic table of `workload`, replayed in roofline.traffic: profiles/rNN_pmc_traffic_<workload>.json,
    or -- config 2 only, the table's original name -- profiles/rNN_pmc_traffic.json.  None when the workload has no table:
    another workload's launches of the same tile are a different problem, their counters are never replayed."""
    pats = ["r[0-9][0-9]_pmc_traffic_%s.json" % workload] + (["r[0-9][0-9]_pmc_traffic.json"] if workload == "cfg2" else [])
    names = sorted(os.path.basename(f) for pat in pats for f in glob.glob(os.path.join(ROOT, "profiles", pat)))
    return names[-1] if names else None


def newest_rocprof_summary(workload="cfg2", f16=False):
    """Newest committed rocprofv3 --kernel-trace --stats summary of this bench command: profiles/rNN_rocprofv3_<workload>_<fp32|f16>_summary.txt."""
    pat = "r[0-9][0-9]_rocprofv3_%s_%s_summary.txt" % (workload.replace("-fp32", ""), "f16" if f16 else "fp32")
    names = sorted(os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", pat)))
    return names[-1] if names else None


def kernel_label(name):
    return ("%s_kernel" if name.startswith(OWN_KERNELS) else "conv_igemm_kernel<%s>") % name


def _tile_key(name):
    """"64x64x32/2x2/m32/dma[N][/chain][/re]" -> (dims + waves + mfma, dma?, stages, chain?, row-major epilogue?)."""
    parts = name.split("/")
    tile, waves, mt = parts[:3]
    rest = parts[3:]
    stage = next((t for t in rest if t.startswith("dma")), "")
    want = [int(v) for v in tile.split("x")] + [int(v) for v in waves.split("x")] + [int(mt[1:])]
    return want, stage.startswith("dma"), (int(stage[3:]) if len(stage) > 3 else 2), "chain" in rest, "re" in rest


def _template_matches(key, name):
    """Does the summarised kernel name `key` (scripts/summarize_prof.py: "conv_igemm<BM,BN,BK,WM,WN,MT,KTAIL,K22,DMA,NSTAGE,
    F16,X3,KWR,CHAIN,REPI>", no blanks, t / f booleans) name the fp32-operand tile configuration `name`?"""
    m = re.match(r"conv_igemm<([^>(]*)", key)
    if not m:
        return False
    targs = [a.strip() for a in m.group(1).split(",")]
    if len(targs) < 10:
        return False

    def flag(a):
        return a in ("t", "true", "1")
    if (len(targs) > 10 and flag(targs[10])) or (len(targs) > 11 and flag(targs[11])):     # fp32-operand tiles only
        return False
    want, want_dma, want_nstage, want_chain, want_re = _tile_key(name)
    chain = len(targs) > 13 and flag(targs[13])
    repi = len(targs) > 14 and flag(targs[14])
    return ([int(a) for a in targs[:6]] == want and flag(targs[8]) == want_dma and int(targs[9]) == want_nstage
            and chain == want_chain and repi == want_re)


def traffic_for(workload, name):
    """(bytes, source) of one kernel from the committed PMC table, or (None, None).  HBM-side traffic is NOT measured by the
    bench command (PMC passes need rocprofv3 around it): the value replayed here comes from the committed summary named in
    `traffic_source`, collected in separate --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command; per dispatch,
    UNCORRECTED (MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads by up to 2x on gfx950, so the true
    figure lies in [traffic, traffic + fetch])."""
    try:
        tfile = newest_traffic_file(workload)
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

        def usable(v):
            return isinstance(v, dict) and v.get("WRITE_SIZE_KiB") is not None and v.get("FETCH_SIZE_KiB") is not None
        if name.startswith(OWN_KERNELS):
            # every template instantiation of that kernel, weighted by its dispatch count: the bench groups them too
            inst = [v for k, v in tj.items() if k.startswith(name + "_kernel") and usable(v)]
            if not inst:
                return None, None
            calls = [float(v.get("calls", 1)) for v in inst]
            mean = {c: sum(v[c] * n for v, n in zip(inst, calls)) / sum(calls) for c in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB")}
            return hit(mean)
        for k, v in tj.items():
            if usable(v) and _template_matches(k, name):
                return hit(v)
    except Exception:
        pass
    return None, None


def rocprof_for(workload, name, f16=False):
    """{"avg_launch_ms", "min_launch_ms", "calls", "file"} of one kernel from the newest committed rocprofv3 kernel-trace
    summary of this bench command, or None.  A profiled pass clocks lower than an un-profiled one (the guide's DVFS note),
    so this is a cross-check of the HIP-event time in the same line, not a replacement."""
    try:
        sfile = newest_rocprof_summary(workload, f16)
        if sfile is None:
            return None
        inst = []
        with open(os.path.join(ROOT, "profiles", sfile)) as f:
            for line in f:
                if line.startswith("## counters"):
                    break
                m = re.match(r"^(\S.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+\d+\s+\d+\s+\d+\s*$", line)
                if not m:
                    continue
                key = m.group(1)
                ok = key.startswith(name + "_kernel") if name.startswith(OWN_KERNELS) else _template_matches(key, name)
                if ok:
                    inst.append((int(m.group(2)), float(m.group(3)), float(m.group(5))))
        if not inst:
            return None
        calls = sum(c for c, _, _ in inst)
        return {"avg_launch_ms": round(sum(t for _, t, _ in inst) / calls / 1e3, 4), "min_launch_ms": round(min(m for _, _, m in inst) / 1e3, 4),
                "calls": calls, "file": os.path.join("profiles", sfile)}
    except Exception:
        return None


def write_rows(path, rows, conv_rows, all_rows):
    """Per-launch detail for tuning sessions (PTX_BENCH_ROWS)."""
    with open(path, "w") as f:
        for (label, macs, ms, cfg, split), stp in zip(rows, [r[6] for r in conv_rows]):
            f.write("%-34s M=%-9d N=%-5d K=%-6d %-28s split=%d %8.4f ms %8.1f TF\n" % (
                label, stp.d.N * stp.d.To * stp.d.Ho * stp.d.Wo, stp.d.Co,
                stp.d.Kc * stp.d.kT * stp.d.kH * stp.d.kW, cfg, split, ms, 2e-9 * macs / ms))
        for lab, kind, nb, macs, ms, _cfg in all_rows:
            if kind == "chain":
                f.write("%-34s chain %-40s %8.4f ms %8.1f TF\n" % (lab, _cfg, ms, 2e-9 * macs / ms))
            elif kind != "conv":
                f.write("%-34s %-5s bytes=%-12d macs=%-14d %8.4f ms %8.1f GB/s\n" % (lab, kind, nb, macs, ms, nb / ms / 1e6))


def kernel_rooflines(eng, model, dev, workload, f16, gflop_per_unit_fixed, units_per_s_per_gpu, ms_per_step, plan=None,
                     single_plan_ms=None):
    """The roofline objects of one bench line from the HIP-event time of every launch of `plan` (default: the newest plan)
    (Engine.profile_steps: an event chain inside ordinary passes, marker overhead calibrated out, so the rows sum to an
    un-instrumented pass).  `ms_per_step`: the headline's step; `single_plan_ms`: the step time of the single-plan
    execution the rows belong to when that is not the headline's (clip lanes), None = not timed.
    Returns (dict of line fields, gflop per unit)."""
    peak_tf = PEAK_F16_MFMA_TF if f16 else PEAK_F32_MFMA_TF
    if plan is None:
        plan = list(eng._plans.values())[-1]
    plan.bind(model)
    with torch.cuda.device(dev):
        all_rows = eng.profile_steps(plan, iters=5)         # EVERY launch of the plan, convs and HBM passes alike
    timing = dict(eng.last_profile or {})
    conv_rows = [r for r in all_rows if r[1] == "conv"]          # (label, kind, bytes, macs, ms, tile, ConvStep)
    all_rows = [r[:6] for r in all_rows]
    rows = [(r[0], r[3], r[4], r[5], r[6].split) for r in conv_rows]
    if os.environ.get("PTX_BENCH_ROWS"):
        write_rows(os.environ["PTX_BENCH_ROWS"], rows, conv_rows, all_rows)
    # the direct kernels (stem, patch-resident 3x3x3) are convs too, and so are the chained launches (two convs in one kernel)
    stem_rows = [(lab, macs, ms, cfg, 1) for lab, kind, nb, macs, ms, cfg in all_rows if kind in ("stem", "chain")]
    by_kernel = {}
    for label, macs, ms, cfg, split in rows + stem_rows:
        k = by_kernel.setdefault(cfg, dict(ms=0.0, flop=0.0, launches=0))
        k["ms"] += ms
        k["flop"] += 2.0 * macs
        k["launches"] += 1
    dom_name, dom = max(by_kernel.items(), key=lambda kv: kv[1]["ms"])
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
    # FLOP the direct kernels actually ISSUE per launch (pruned temporal taps excluded, K padded 21 -> 22): the
    # host-side twin of SQ_INSTS_MFMA x 4096 from the committed PMC pass
    issued_by_kernel = {}
    for stp in plan.steps:
        for t in (stp.active() if hasattr(stp, "active") else [stp]):
            if type(t).__name__ == "StemF32Step" or getattr(t, "body", None) is not None:
                issued_by_kernel[t.kernel] = issued_by_kernel.get(t.kernel, 0.0) + t.issued_flop()

    def roof(name, ms, flop, launches):
        traffic, traffic_source = traffic_for(workload, name)
        tf = flop / (ms * 1e-3) / 1e12
        r = {"bound": "mfma", "kernel": kernel_label(name),
             "achieved": round(tf, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(tf / peak_tf, 4),
             "traffic": traffic, "traffic_source": traffic_source, "launches_per_step": launches,
             "avg_launch_ms": round(ms / launches, 4), "algorithmic_gflop_per_launch": round(flop / launches / 1e9, 3)}
        # the same kernel in the committed rocprofv3 summary of this command (cross-check): valid when the
        # profile has a whole number of steps' worth of dispatches of it
        rp = rocprof_for(workload, name, f16)
        if rp and rp["calls"] % launches == 0:
            rp["frac_rocprof"] = round(flop / launches / (rp["avg_launch_ms"] * 1e-3) / 1e12 / peak_tf, 4)
            rp["hip_event_over_rocprof"] = round(ms / launches / rp["avg_launch_ms"], 4)
        r["rocprof"] = rp
        r["frac_rocprof"] = rp.get("frac_rocprof") if rp else None
        # `frac` prices padding taps as work (SURVEY.md 8d allows it); `issued_frac` = MFMA FLOP the kernel really
        # issues / time / peak -- null for the generic tiles, whose tap pruning is decided per tile at run time
        iss = issued_by_kernel.get(name)
        r["issued_gflop_per_launch"] = round(iss / launches / 1e9, 3) if iss else None
        r["issued_frac"] = round(iss / (ms * 1e-3) / 1e12 / peak_tf, 4) if iss else None
        return r

    # `roofline`: the kernel (one template instantiation, as rocprofv3 --stats groups them) with the largest total time
    # per step; `roofline_longest_launch`: the single longest launch of the step (the stem), whose average duration is the
    # one-problem row of the committed rocprofv3 summary
    roofline = roof(dom_name, dom["ms"], dom["flop"], dom["launches"])
    ll = max(rows + stem_rows, key=lambda r: r[2])
    roofline_longest = roof(ll[3], ll[2], 2.0 * ll[1], 1)
    roofline_longest["label"] = ll[0]
    gflop_per_unit = gflop_per_unit_fixed if gflop_per_unit_fixed else sum(2e-9 * r[1] for r in rows + stem_rows) / plan.shape[0]
    net_tf = gflop_per_unit * 1e9 * units_per_s_per_gpu / 1e12
    # `frac` prices padding taps as work (SURVEY.md 8d's convention: layer4's T = 1 3x3x3 convs then "run" above the
    # peak); `issued_frac` is its twin on the FLOP the MFMA instructions of one step really issue (pruned tap planes
    # excluded, tile / K padding included: engine.issued_conv_flop, the stem's issued_flop) over the same step time
    issued_step = sum(t.issued_flop() for t in plan.all_convs() if hasattr(t, "issued_flop"))
    issued_tf = issued_step / (ms_per_step * 1e-3) / 1e12 if not f16 else None
    non_conv_ms = sum(v["ms"] for v in roofline_hbm.values()) + other_ms
    roofline_net = {"bound": "mfma", "achieved": round(net_tf, 2), "peak": peak_tf,
                    "unit": "TFLOP/s", "frac": round(net_tf / peak_tf, 4),
                    "issued_gflop_per_step": round(issued_step / 1e9, 3) if issued_tf is not None else None,
                    "issued_frac": round(issued_tf / peak_tf, 4) if issued_tf is not None else None,
                    "conv_ms_sum": round(conv_ms, 3),
                    "per_kernel": {k: {"ms": round(v["ms"], 3), "tflops": round(v["flop"] / v["ms"] / 1e9, 1),
                                       "launches": v["launches"]} for k, v in sorted(by_kernel.items())}}
    # the invariant VERDICT r5 #1 asks for: the launches of a step cannot take longer than the step
    timing = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in timing.items()}
    step_of_rows = single_plan_ms if single_plan_ms is not None else (timing.get("plain_pass_ms") or ms_per_step)
    timing["launch_ms_sum"] = round(conv_ms + non_conv_ms, 4)
    timing["ms_per_step"] = round(step_of_rows, 4)
    timing["launch_sum_le_step"] = bool(conv_ms + non_conv_ms <= step_of_rows * 1.01)
    fields = {"roofline": roofline, "roofline_longest_launch": roofline_longest, "roofline_net": roofline_net,
              "roofline_hbm": roofline_hbm, "non_conv_ms": round(non_conv_ms, 4), "launch_timing": timing}
    return fields, gflop_per_unit


def _parity(got, want, tolerance, **extra):
    d = {"max_abs_dlogits": float((got - want).abs().max().item()), "max_abs_logit": float(want.abs().max().item()),
         "argmax_equal": bool(torch.equal(got.argmax(1), want.argmax(1))) if got.dim() == 2 else None, "tolerance": tolerance}
    d.update(extra)
    return d


def shard_parity(run, cpu_fn, sd, x_cpu, idx, tolerance, world, unit):
    """N > 1: the CPU baseline is reported at N = 1 only (bench contract), but the line stays self-verifying -- rank 0
    checks ITS OWN shard against the oracle (one bounded CPU forward)."""
    xs = x_cpu[idx]
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 1) // max(1, world))))
    want = cpu_fn(sd, xs, idx)
    got = run().cpu()[idx]
    return _parity(got, want, tolerance, scope="rank 0's shard (%d %s) vs the CPU oracle" % (xs.shape[0], unit)), want


def cpu_baseline_leg(run, cpu_fn, sd, x_cpu, idx, tolerance, headline, units_per_gpu, unit):
    """The oracle restatement of the reference path timed on this box's host cores (bounded to ~30 s) + the parity of the
    HIP path against it.  Returns (cpu_baseline dict, parity dict, oracle output)."""
    ncpu = os.cpu_count() or 1
    xs = x_cpu[idx]
    # pick the thread count that runs the reference path fastest on this host (SMT oversubscription makes oneDNN conv3d
    # collapse), then time it
    cands = sorted({c for c in (16, 32, 64, 128, ncpu // 2) if 1 <= c <= ncpu}) or [ncpu]
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
    return cpu, _parity(got, want, tolerance, units_checked=[int(i) for i in idx]), want


def _timed(run, steps, warmup):
    out = None
    for _ in range(warmup):
        out = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


def x3_leg(args, eng, model, x, run, dev, headline, full_tune, units_per_gpu, unit, gflop_per_unit, fp32_rate, want, idx, tolerance):
    """Secondary leg: the same workload with Engine.precision = "x3" (fp32 operands split into half pairs, three fp16
    MFMAs per product block, fp32 accumulate -- fp32-ACCURATE).  Reported next to the headline with its own
    |d output| vs the CPU path and its own denominator (the fp16 dense MFMA peak / 3 issued MFMAs per algorithmic
    product); the headline `value` stays the plain fp32-MFMA path."""
    eng.precision = "x3"
    if not args.no_autotune:
        if headline or full_tune:
            eng.autotune(model, x, iters=2, verbose=args.verbose)
        else:
            run()
    el3, out3 = _timed(run, args.steps, args.warmup)
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
    byk = {}
    # the direct split-operand stem (ptx_conv_stem_x3_fwd) is a conv too, with its own kernel
    for lab, kind, nb, macs, ms, cfg in [r for r in rows3 if r[1] in ("conv", "stem")]:
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
             "speedup_vs_fp32_mfma": round(rate3 / fp32_rate, 3),
             "roofline": {"bound": "mfma", "kernel": kernel_label(dn),
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
    if want is not None:
        got3 = out3.cpu()[idx]
        split["parity"] = {"max_abs_dlogits": float((got3 - want).abs().max().item()),
                           "argmax_equal": bool(torch.equal(got3.argmax(1), want.argmax(1))) if got3.dim() == 2 else None,
                           "tolerance": tolerance}
    eng.precision = "fp32"
    return split


def lanes_leg(args, eng, model, x, run, headline, full_tune, units_per_gpu, unit, other_rate, other_label, out_ref, want, idx, tolerance,
              split, lanes):
    """Secondary leg: the same batch with Engine.lanes = `lanes` (2 = two half-batch plans on two HIP
    streams, 1 = the single-plan path) -- whichever the headline did NOT run, so both figures are in every line."""
    keep = eng.lanes
    eng.lanes = lanes
    part = x[:units_per_gpu // lanes]
    tune = not args.no_autotune and (headline or full_tune)
    if tune and lanes > 1:
        eng.autotune(model, part, iters=int(os.environ.get("PTX_TUNE_ITERS", "2")), verbose=args.verbose)
    el_l, out_l = _timed(run, args.steps, max(args.warmup, 1))
    rate_l = units_per_gpu * args.steps / el_l
    leg = {"lanes": lanes, "value": round(rate_l, 2), "unit": "%s/s" % unit, "ms_per_step": round(1e3 * el_l / args.steps, 4),
           "speedup_vs_%s" % other_label: round(rate_l / other_rate, 4),
           "launch_shape": "%d %s per launch (%d plan(s) of %d-%s batches, own buffers, tiles tuned for that shape)" % (
               units_per_gpu // lanes, unit, lanes, units_per_gpu // lanes, unit[:-1]),
           "max_abs_d_vs_headline_path": float((out_l - out_ref).abs().max().item()),
           "argmax_equal_headline_path": bool(torch.equal(out_l.argmax(1), out_ref.argmax(1))) if out_l.dim() == 2 else None,
           "parity": None}
    if want is not None:
        got_l = out_l.cpu()[idx]
        leg["parity"] = {"max_abs_dlogits": float((got_l - want).abs().max().item()),
                         "argmax_equal": bool(torch.equal(got_l.argmax(1), want.argmax(1))) if got_l.dim() == 2 else None,
                         "tolerance": tolerance}
    if split is not None and lanes > 1:
        # ... and the split-operand arithmetic through the same lanes (its shorter matrix work leaves more of a launch to
        # gaps and HBM-bound passes, so the overlap pays more)
        eng.precision = "x3"
        if tune:
            eng.autotune(model, part, iters=int(os.environ.get("PTX_TUNE_ITERS", "2")), verbose=args.verbose)
        el_l3, out_l3 = _timed(run, args.steps, max(args.warmup, 1))
        rate_l3 = units_per_gpu * args.steps / el_l3
        leg["split_f16x3"] = {"value": round(rate_l3, 2), "unit": "%s/s" % unit, "ms_per_step": round(1e3 * el_l3 / args.steps, 4),
                              "speedup_vs_single_plan_x3": round(rate_l3 / split["value"], 4), "parity": None}
        if want is not None:
            got_l3 = out_l3.cpu()[idx]
            leg["split_f16x3"]["parity"] = {
                "max_abs_dlogits": float((got_l3 - want).abs().max().item()),
                "argmax_equal": bool(torch.equal(got_l3.argmax(1), want.argmax(1))) if got_l3.dim() == 2 else None,
                "tolerance": tolerance}
        eng.precision = "fp32"
    eng.lanes = keep
    return leg
