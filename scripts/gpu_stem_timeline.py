"""Phase clock of the direct fp32 stem (the headline's dominant kernel): per-workgroup medians of
issue (entry -> first requests out) | land (-> first patch in LDS) | steps (-> last (kt, kh) step) | store (-> stores retired),
workgroups per CU and how much of a CU's span has >= 1 / >= 2 / >= 3 of its resident workgroups inside the step loop.
Diagnostic library: scripts/micro/build_timeline.sh.

    python scripts/gpu_stem_timeline.py resnet3d50 8x3x16x224x224
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd import engine as E  # noqa: E402
from pretorched_x_amd.testing import synth_clips, synth_state_dict  # noqa: E402

L = ptx._lib
L.LIB_PATH = os.path.join(ROOT, "scripts", "micro", "libptx_amd_tl.so")
lib = L.lib()
lib.ptx_stem_f32_timeline.restype = C.c_int
lib.ptx_stem_f32_timeline.argtypes = [C.c_void_p]

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet3d50"
shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8x3x16x224x224").split("x"))
kw = dict(num_classes=339, pretrained=None) if arch == "resnet3d50" else dict(pretrained=None) if arch == "nonlocalresnet3d50" else dict(num_classes=339)
DEV = "cuda:0"
m = ptx.__dict__[arch](**kw)
m.load_state_dict(synth_state_dict(m.state_dict(), 1234))
m = m.to(DEV).eval()
x = synth_clips(shape[0], shape[2], shape[3], 99).to(DEV)
with torch.no_grad():
    for _ in range(3):
        m(x)
torch.cuda.synchronize()
plan = next(iter(m.engine()._plans.values()))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
stem = next(s for s in plan.steps if isinstance(s, E.StemF32Step))
for _ in range(20):                               # warm clocks
    stem(st)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    stem(st)
e1.record()
torch.cuda.synchronize()
print("%s %s: stem launch %.4f ms (HIP events, 10 launches)" % (arch, "x".join(map(str, shape)), e0.elapsed_time(e1) / 10))
tl = torch.zeros(1 << 17, 8, dtype=torch.int64, device=DEV)
L.check(lib.ptx_stem_f32_timeline(C.c_void_p(tl.data_ptr())), "on")
for _ in range(5):
    stem(st)                                       # the last launch's clocks stay in the buffer
torch.cuda.synchronize()
L.check(lib.ptx_stem_f32_timeline(None), "off")
t = tl.cpu().numpy()
t = t[t[:, 0] > 0]
tick = 0.01
names = ["issue", "land", "steps", "store"]
cols = [(0, 1), (1, 2), (2, 3), (3, 5)]
print("%d workgroups; per-workgroup phases, us (mean / median / p90):" % len(t))
for nm, (a, b) in zip(names, cols):
    d = (t[:, b] - t[:, a]) * tick
    print("  %-6s %7.2f / %7.2f / %7.2f" % (nm, d.mean(), np.median(d), np.percentile(d, 90)))
life = (t[:, 5] - t[:, 0]) * tick
print("  life   %7.2f / %7.2f / %7.2f   (short workgroups = output frames with fewer valid temporal taps)" % (life.mean(), np.median(life), np.percentile(life, 90)))
print("first entry -> last exit: %.1f us" % ((t[:, 5].max() - t[:, 0].min()) * tick))
cu = t[:, 6]
ids = np.unique(cu)
per = np.array([(cu == i).sum() for i in ids])
print("%d distinct (xcc, se, cu) ids; workgroups per id min / mean / max = %d / %.1f / %d" % (len(ids), per.min(), per.mean(), per.max()))
occ = np.zeros(5)
spans, ends = [], []
for i in ids[:: max(1, len(ids) // 64)]:
    rows = t[cu == i]
    ev = sorted([(r[2], 1) for r in rows] + [(r[3], -1) for r in rows])
    lo, hi = rows[:, 0].min(), rows[:, 5].max()
    cur, last = 0, lo
    acc = np.zeros(5)
    for tt, dlt in ev:
        acc[min(cur, 4)] += tt - last
        cur += dlt
        last = tt
    acc[0] += hi - last
    occ += acc / max(1, hi - lo)
    spans.append((hi - lo) * tick)
    ends.append((hi - t[:, 0].min()) * tick)
occ /= len(spans)
print("sampled CUs: share of the CU's span with 0 / 1 / 2 / 3 / 4+ workgroups inside the step loop: " + " / ".join("%.0f %%" % (100 * v) for v in occ))
print("CU busy span %.1f .. %.1f us; the CUs finish between %.1f and %.1f us after the first entry" % (min(spans), max(spans), min(ends), max(ends)))
