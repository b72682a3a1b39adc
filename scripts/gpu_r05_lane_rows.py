"""Where do half-batch launches lose against full-batch ones?  Per-layer sums of the in-sequence launch times of the
8-clip plan and of the 4-clip plan (x 2), config 2.  Measurement only."""
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import synth_clips, synth_state_dict  # noqa: E402

m = ptx.resnet3d50(num_classes=339, pretrained=None)
m.load_state_dict(synth_state_dict(m.state_dict(), 1234))
m = m.cuda().eval()
x = synth_clips(8, 16, 224, 99).cuda()
eng = m.engine()
sums = {}
for n in (8, 4):
    xb = x[:n].contiguous()
    eng.autotune(m, xb, iters=int(os.environ.get("PTX_TUNE_ITERS", "4")))
    m(xb)
    plan = eng.lane_plans(m, xb)[0]
    plan.bind(m)
    rows = eng.profile_steps(plan, iters=8)
    g = collections.OrderedDict()
    for r in rows:
        lab = r[0]
        key = lab.split(".")[0] if lab.startswith("layer") else ("stem" if lab.startswith("conv1") else lab.split(".")[0])
        g[key] = g.get(key, 0.0) + r[4]
    sums[n] = g
print("%-12s %10s %10s %8s" % ("group", "8 clips ms", "2x4 ms", "ratio"))
for k in sums[8]:
    a, b = sums[8][k], 2 * sums[4].get(k, 0.0)
    print("%-12s %10.4f %10.4f %8.3f" % (k, a, b, b / a if a else 0))
print("%-12s %10.4f %10.4f" % ("total", sum(sums[8].values()), 2 * sum(sums[4].values())))
