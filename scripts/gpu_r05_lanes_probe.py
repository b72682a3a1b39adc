"""Clip lanes probe: does a batch run faster as TWO half-batch forwards on two HIP streams (each chain filling the other's
launch gaps and tile tails) than as one forward?  Whole network, eager launches, tiles tuned for the half-batch shapes.

    python scripts/gpu_r05_lanes_probe.py cfg2|cfg3 [steps]

Prints clips/s for: one forward of the full batch | two half-batch forwards on ONE stream | on TWO streams | 4 quarter-batch
forwards on four streams.  Measurement only -- nothing here is product code.
"""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import synth_clips, synth_state_dict  # noqa: E402


def build(workload):
    if workload == "cfg2":
        m = ptx.resnet3d50(num_classes=339, pretrained=None)
        sd = synth_state_dict(m.state_dict(), 1234)
        x = synth_clips(8, 16, 224, 99)
    else:
        m, recipe, make, per_gpu, _fwd, _cpu, _unit, _label, _idx = bench.other_workload(workload, 0)
        sd = synth_state_dict(m.state_dict(), 1234, **recipe)
        x = make(per_gpu, 99)
    m.load_state_dict(sd)
    return m.cuda().eval(), x.cuda()


def timed(fn, steps, warmup=5):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    iters = int(os.environ.get("PTX_TUNE_ITERS", "3"))
    model, x = build(workload)
    n = x.shape[0]
    with torch.no_grad():
        ref = model(x).clone()
        t_full = timed(lambda: model(x), steps)
        print("%s full batch %d, one forward            %8.1f clips/s  %.4f ms" % (workload, n, n / t_full, 1e3 * t_full), flush=True)
        for lanes in (2, 4):
            if n % lanes:
                continue
            per = n // lanes
            parts = [x[i * per:(i + 1) * per].contiguous() for i in range(lanes)]
            models = [model] + [copy.deepcopy(model) for _ in range(lanes - 1)]       # own engine, own plans, own buffers
            t0 = time.perf_counter()
            models[0].engine().autotune(models[0], parts[0], iters=iters)             # the table is global: the copies reuse it
            for m in models[1:]:
                m.engine().invalidate()
            torch.cuda.synchronize()
            print("   tuned the %d-clip shapes in %.1f s" % (per, time.perf_counter() - t0), flush=True)
            streams = [torch.cuda.Stream() for _ in range(lanes)]
            outs = [None] * lanes

            def serial():
                for i in range(lanes):
                    outs[i] = models[i](parts[i])

            def overlapped():
                cur = torch.cuda.current_stream()
                for i in range(lanes):
                    streams[i].wait_stream(cur)
                    with torch.cuda.stream(streams[i]):
                        outs[i] = models[i](parts[i])
                for i in range(lanes):
                    cur.wait_stream(streams[i])

            def free_running():          # no per-step join: the upper bound of the overlap (the lanes drift apart)
                for i in range(lanes):
                    with torch.cuda.stream(streams[i]):
                        outs[i] = models[i](parts[i])

            t_ser = timed(serial, steps)
            got = torch.cat(outs)
            t_ovl = timed(overlapped, steps)
            got2 = torch.cat(outs)
            print("%s %d x %d clips, one stream              %8.1f clips/s  %.4f ms   max|d| vs full %.2e" % (
                workload, lanes, per, n / t_ser, 1e3 * t_ser, (got - ref).abs().max().item()), flush=True)
            print("%s %d x %d clips, %d streams               %8.1f clips/s  %.4f ms   max|d| vs full %.2e  (%.3fx the full batch)" % (
                workload, lanes, per, lanes, n / t_ovl, 1e3 * t_ovl, (got2 - ref).abs().max().item(), t_full / t_ovl), flush=True)
            for rep in range(2):
                t_o = timed(overlapped, steps)
                t_f = timed(free_running, steps)
                t_1 = timed(lambda: model(x), steps)
                print("   repeat %d: %d streams %8.1f | free-running %8.1f | full batch %8.1f clips/s   (%.3fx / %.3fx)" % (
                    rep, lanes, n / t_o, n / t_f, n / t_1, t_1 / t_o, t_1 / t_f), flush=True)
        t_full2 = timed(lambda: model(x), steps)
        print("%s full batch again                      %8.1f clips/s  %.4f ms" % (workload, n / t_full2, 1e3 * t_full2), flush=True)


if __name__ == "__main__":
    main()
