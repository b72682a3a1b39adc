"""Round-6 probe: two clip lanes of UNEQUAL size.  Equal lanes run the same kernel sequence in near lock step (both in the
MFMA-bound stem together, both in the HBM-bound pointwise convs together); unequal slices de-phase the two chains.  Whole
network, eager launches, tiles tuned per slice shape.    python scripts/gpu_r06_uneven_lanes_probe.py cfg2|cfg3 [steps]"""
import copy
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import gpu_r05_lanes_probe as P  # noqa: E402


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    iters = int(os.environ.get("PTX_TUNE_ITERS", "3"))
    model, x = P.build(workload)
    model.engine().lanes = 1
    n = x.shape[0]
    other = copy.deepcopy(model)
    other.engine().lanes = 1
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.no_grad():
        t_full = P.timed(lambda: model(x), steps)
        print("%s full batch %d, one plan        %8.1f clips/s  %.4f ms" % (workload, n, n / t_full, 1e3 * t_full), flush=True)
        for a in (n // 2, n // 2 + 1, n // 2 + 2):
            pa, pb = x[:a].contiguous(), x[a:].contiguous()
            model.engine().autotune(model, pa, iters=iters)
            model.engine().autotune(model, pb, iters=iters)
            other.engine().invalidate()
            outs = [None, None]

            def overlapped():
                cur = torch.cuda.current_stream()
                s0.wait_stream(cur)
                s1.wait_stream(cur)
                with torch.cuda.stream(s0):
                    outs[0] = model(pa)
                with torch.cuda.stream(s1):
                    outs[1] = other(pb)
                cur.wait_stream(s0)
                cur.wait_stream(s1)
            for rep in range(3):
                t_o = P.timed(overlapped, steps)
                t_1 = P.timed(lambda: model(x), steps)
                print("   %d + %d clips on two streams  %8.1f clips/s | full batch %8.1f   (%.3fx)" % (a, n - a, n / t_o, n / t_1, t_1 / t_o), flush=True)


if __name__ == "__main__":
    main()
