"""Where does a workgroup of the generic implicit-GEMM kernel spend its time on the SMALL launches (layer3 / layer4 of config
2, most of config 3)?  The per-launch model t = t0 + FLOP / R puts 8-13 us of every launch into t0; this prints
what t0 is made of, per conv step, from thread 0's 100 MHz wall clock (diagnostic library: scripts/micro/build_timeline.sh):

  setup (entry -> operand tables built) | first (-> first tile landed) | loop (-> last k-step) | epi (-> stores retired)
  plus the launch's span (first entry -> last exit), the HIP-event time of the same launch and workgroups per CU.

    python scripts/gpu_igemm_timeline.py resnet3d50 8x3x16x224x224 layer3.1 layer4.1 layer2.1
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
lib.ptx_igemm_timeline.restype = C.c_int
lib.ptx_igemm_timeline.argtypes = [C.c_void_p]

arch = sys.argv[1] if len(sys.argv) > 1 else "resnet3d50"
shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8x3x16x224x224").split("x"))
pats = sys.argv[3:] or ["layer3.1", "layer4.1"]
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
eng = m.engine()
plan = next(iter(eng._plans.values()))
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
flat = [t for s in plan.steps for t in (s.active() if isinstance(s, E.AltStep) else [s])]
MAXWG = 1 << 18
tl = torch.zeros(MAXWG, 8, dtype=torch.int64, device=DEV)
tick = 0.01
print("%-34s %-30s %5s %5s | %6s %6s %6s %6s | %6s %6s %6s" % ("step", "tile", "wgs", "wg/cu", "setup", "first", "loop", "epi", "life", "span", "event"))
for stp in flat:
    if not isinstance(stp, E.ConvStep) or not any(p in stp.label for p in pats):
        continue
    for _ in range(3):
        stp(st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        stp(st)
    e1.record()
    torch.cuda.synchronize()
    ev = e0.elapsed_time(e1) / 10 * 1000
    tl.zero_()
    L.check(lib.ptx_igemm_timeline(C.c_void_p(tl.data_ptr())), "on")
    stp(st)
    torch.cuda.synchronize()
    L.check(lib.ptx_igemm_timeline(None), "off")
    t = tl.cpu().numpy()
    t = t[t[:, 0] > 0]
    if not len(t):
        print("%-34s no workgroups reported" % stp.label)
        continue
    # split-K launches add a reduce kernel (not traced); the row-major epilogue reports 4, the column-wise one does not
    have4 = (t[:, 4] > 0).all()
    setup = (t[:, 1] - t[:, 0]) * tick
    first = (t[:, 2] - t[:, 1]) * tick
    loop = (t[:, 3] - t[:, 2]) * tick
    epi = (t[:, 5] - t[:, 3]) * tick
    life = (t[:, 5] - t[:, 0]) * tick
    span = (t[:, 5].max() - t[:, 0].min()) * tick
    ncu = len(np.unique(t[:, 6]))
    name = lib.ptx_conv3d_config_name(stp.cfg).decode()
    print("%-34s %-30s %5d %5.2f | %6.2f %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f%s" % (
        stp.label, name + ("/s%d" % stp.split if stp.split > 1 else ""), len(t), len(t) / ncu, np.median(setup), np.median(first), np.median(loop), np.median(epi),
        np.median(life), span, ev, "" if have4 else "  (column-wise epilogue)"))
