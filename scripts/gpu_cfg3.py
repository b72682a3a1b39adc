"""Timing of BASELINE.json config 3 (8x3x32x112x112): (2+1)D + non-local composite and parents."""
import sys
import time

import torch

sys.path.insert(0, ".")
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import synth_clips, synth_state_dict  # noqa: E402

GFLOP_PER_CLIP = {"nonlocal_r2plus1d50": 53.343, "r2plus1d50": 42.32, "nonlocalresnet3d50": 50.1, "resnet3d50": None}
for arch, kw, recipe in [("nonlocal_r2plus1d50", dict(num_classes=339), dict(inner_bn_damp=0.9, nl_bn_damp=0.05)),
                         ("r2plus1d50", dict(num_classes=400), dict(inner_bn_damp=0.9)),
                         ("nonlocalresnet3d50", dict(pretrained=None), dict(last_bn_damp=0.65, nl_bn_damp=0.05))]:
    m = ptx.__dict__[arch](**kw)
    m.load_state_dict(synth_state_dict(m.state_dict(), 1234, **recipe))
    m = m.cuda().eval()
    x = synth_clips(8, 32, 112, 99).cuda()
    t0 = time.time()
    m.engine().autotune(m, x, iters=2)
    ttune = time.time() - t0
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        y = m(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    gf = GFLOP_PER_CLIP[arch]
    print("%-22s %8.3f ms/step  %8.1f clips/s  %6.1f TF (%.1f%% of 157.3)  [autotune %.0f s]" % (
        arch, ms, 8e3 / ms, gf * 8 / ms, gf * 8 / ms / 1.573, ttune), flush=True)
    rows = m.engine().profile_convs(m, x, iters=3)
    tot = sum(r[2] for r in rows)
    worst = sorted(rows, key=lambda r: -r[2])[:6]
    print("   conv launches %d, conv time %.3f ms; slowest:" % (len(rows), tot))
    for lab, macs, t, cfg, sk in worst:
        print("     %-40s %.3f ms  %6.1f TF  %s split=%d" % (lab, t, 2e-9 * macs / t, cfg, sk))
