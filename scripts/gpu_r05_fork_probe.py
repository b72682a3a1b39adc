"""Would forking into clip lanes only for the TAIL of the network beat whole-network lanes?  Times step ranges of the
full-batch plan and of the two half-batch lane plans (overlapped on two streams), config 2 / 3:
    hybrid estimate(cut) = full-batch steps [0, cut) + two lanes' steps [cut, end) overlapped.
Measurement only (drives Plan.steps directly)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd import engine as E  # noqa: E402
from gpu_r05_lanes_probe import build  # noqa: E402


def label(s):
    t = s.active()[0] if hasattr(s, "active") else s
    return getattr(t, "label", "") or ""


def cut_index(plan, prefix):
    for i, s in enumerate(plan.steps):
        if label(s).startswith(prefix):
            return i
    return len(plan.steps)


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    model, x = build(workload)
    eng = model.engine()
    n = x.shape[0]
    with torch.no_grad():
        eng.autotune(model, x, iters=4)
        eng.autotune(model, x[:n // 2].contiguous(), iters=4)
        model(x)
        p8 = eng.lane_plans(model, x)[0]
        eng.lanes = 2
        model(x)
        p4 = eng.lane_plans(model, x)
        eng.lanes = 1
    halves = [x[:n // 2].contiguous(), x[n // 2:].contiguous()]
    for p in [p8] + p4:
        p.bind(model)
    p8.in_ptr = E._ptr(x)
    for p, h in zip(p4, halves):
        p.in_ptr = E._ptr(h)
    side = torch.cuda.Stream()

    def run(plan, lo, hi):
        st = E._stream()
        for s in plan.steps[lo:hi]:
            s(st)

    def full(lo, hi):
        run(p8, lo, hi)

    def lanes(lo4, hi4):
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            run(p4[1], lo4[1], hi4[1])
        run(p4[0], lo4[0], hi4[0])
        cur.wait_stream(side)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / reps

    n8 = len(p8.steps)
    n4 = [len(p.steps) for p in p4]
    t_full = timed(lambda: full(0, n8))
    t_lanes = timed(lambda: lanes([0, 0], n4))
    print("%s: %d steps (full-batch plan), %s steps (lane plans)   all steps: one plan %.4f ms | two lanes %.4f ms" % (
        workload, n8, n4, t_full, t_lanes), flush=True)
    print("%-10s %12s %12s %12s %12s %14s" % ("cut at", "full[0,cut)", "full[cut,)", "lanes[0,cut)", "lanes[cut,)", "hybrid estimate"))
    for cut in ("layer1.0", "layer2.0", "layer3.0", "layer3.3", "layer4.0"):
        c8 = cut_index(p8, cut)
        c4 = [cut_index(p, cut) for p in p4]
        a = timed(lambda: full(0, c8))
        b = timed(lambda: full(c8, n8))
        c = timed(lambda: lanes([0, 0], c4))
        d = timed(lambda: lanes(c4, n4))
        print("%-10s %12.4f %12.4f %12.4f %12.4f %14.4f   (%.3fx one plan)" % (cut, a, b, c, d, a + d, t_full / (a + d)), flush=True)


if __name__ == "__main__":
    main()
