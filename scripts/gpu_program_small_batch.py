"""Conv programs at SMALL batches (the strong-scaling regime: 8 clips over 8 GPUs = 1 clip per GPU): resnet3d50 16 x 224^2 and
the config-3 composite 32 x 112^2 at 1 / 2 / 4 clips, PTX_PROGRAM=0 vs force (one process per setting: the choice is made at
plan build).  Usage: python scripts/gpu_program_small_batch.py <mode> ; prints ms per forward."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PTX_PROGRAM"] = sys.argv[1]
import torch  # noqa: E402

import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import synth_state_dict  # noqa: E402

for name, kw, shape in (("resnet3d50", dict(num_classes=339, pretrained=None), (3, 16, 224, 224)),
                        ("nonlocal_r2plus1d50", dict(num_classes=339), (3, 32, 112, 112))):
    m = ptx.__dict__[name](**kw)
    m.load_state_dict(synth_state_dict(m.state_dict(), 1234))
    m = m.cuda().eval()
    for b in (1, 2, 4):
        x = torch.randn(b, *shape, device="cuda")
        with torch.no_grad():
            for _ in range(3):
                y = m(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                y = m(x)
            torch.cuda.synchronize()
        plan = list(m.engine()._plans.values())[-1]
        print("%-22s batch %d  PTX_PROGRAM=%-5s %8.3f ms  (%d conv launches, %d programs)" % (
            name, b, sys.argv[1], 1e3 * (time.perf_counter() - t0) / 20, len(plan.all_convs()), sum(p.use_program for p in plan.program_steps)),
            flush=True)
