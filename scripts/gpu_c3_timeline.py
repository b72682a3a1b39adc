"""Where does a workgroup of ptx_conv3x3_f16_fwd spend its time?  (round 4: three scheduling experiments on the kernel were
neutral -- counted waits, two taps per barrier, three workgroups per CU -- so measure the phases instead of guessing.)

Runs the kernel from the DIAGNOSTIC library (scripts/micro/build_timeline.sh: the same sources with -DPTX_C3_TIMELINE) on one
generator-stage shape and prints, from thread 0's 100 MHz wall clock per workgroup:
  setup (entry -> requests issued) | load (-> patch + first filter tile landed) | taps (-> last MFMA issued) |
  park (-> output tile in LDS) | store (-> stores retired), plus how many workgroups a CU ran and its busy span.

    python scripts/gpu_c3_timeline.py 64 256 256 64 [up2]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pretorched_x_amd as ptx  # noqa: E402

L = ptx._lib
L.LIB_PATH = os.path.join(ROOT, "scripts", "micro", "libptx_amd_tl.so")
lib = L.lib()
lib.ptx_c3_timeline.restype = C.c_int
lib.ptx_c3_timeline.argtypes = [C.c_void_p]

N, H, W, Cc = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (64, 256, 256, 64)
up2 = len(sys.argv) > 5 and sys.argv[5] == "up2"
DEV = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
Hs, Ws = (H // 2, W // 2) if up2 else (H, W)
g = torch.Generator().manual_seed(1)
x = (torch.randn(N, 1, Hs, Ws, Cc, generator=g) * 0.5).half().to(DEV)
w = (torch.randn(Cc, Cc, 1, 3, 3, generator=g) * (Cc * 9) ** -0.5).to(DEV)
b = torch.randn(Cc, generator=g).to(DEV)
pd = L.PackDesc(Cc, Cc, 1, 3, 3, Cc, (Cc + 127) // 128 * 128, 0, 0, 0, 0, 0, 0, 1)
wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV, dtype=torch.float16)
bp = torch.empty(pd.Co_pad, device=DEV)
L.check(lib.ptx_pack_conv_weight(C.byref(pd), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), None, None, None, None,
                                 C.c_float(0), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()), st), "pack")
d = L.ConvDesc()
d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, 1, H, W, Cc // 2, Cc // 2
d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, H, W, Cc, Cc
d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 1, 3, 3, 1, 1, 1, 0, 1, 1
d.Kc, d.Co_pad, d.groups = Cc // 2, pd.Co_pad, 1
d.flags = L.PTX_F16_OPERANDS | L.PTX_EPI_OUT_F16 | L.PTX_EPI_AFFINE | L.PTX_EPI_RELU | (L.PTX_PRO_UP2 if up2 else 0)
assert lib.ptx_conv3x3_f16_supported(C.byref(d))
sc = (torch.rand(N, Cc) + 0.5).to(DEV)
sh = (torch.randn(N, Cc) * 0.3).to(DEV)
ext = L.ConvFusedExt()
ext.scale, ext.shift, ext.ld_affine = sc.data_ptr(), sh.data_ptr(), Cc
y = torch.empty(N, 1, H, W, Cc, device=DEV, dtype=torch.float16)


def run():
    L.check(lib.ptx_conv3x3_f16_fwd(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(wp.data_ptr()), C.c_void_p(bp.data_ptr()),
                                    C.c_void_p(y.data_ptr()), C.byref(ext), st), "conv3x3_f16")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
print("kernel without the clock writes: %.4f ms per launch" % (e0.elapsed_time(e1) / 10))

parts = 2 if Cc == 256 else 1
n_wg = N * ((H + 7) // 8) * ((W + 31) // 32) * parts
tl = torch.zeros(n_wg, 8, dtype=torch.int64, device=DEV)
L.check(lib.ptx_c3_timeline(C.c_void_p(tl.data_ptr())), "timeline on")
e0.record()
run()
e1.record()
torch.cuda.synchronize()
L.check(lib.ptx_c3_timeline(None), "timeline off")
print("instrumented launch: %.4f ms, %d workgroups" % (e0.elapsed_time(e1), n_wg))
t = tl.cpu().numpy().astype(np.int64)
assert (t[:, 0] > 0).all(), "some workgroups did not report"
tick = 0.01                                           # us per tick of the 100 MHz clock
ph = np.diff(t[:, :6], axis=1) * tick                 # [wg][setup, load, taps, park, store]
names = ["setup", "load", "taps", "park", "store"]
print("per-workgroup phases, us (mean / median / p90):")
for k, nm in enumerate(names):
    print("  %-6s %6.2f / %6.2f / %6.2f" % (nm, ph[:, k].mean(), np.median(ph[:, k]), np.percentile(ph[:, k], 90)))
life = (t[:, 5] - t[:, 0]) * tick
print("  life   %6.2f / %6.2f / %6.2f" % (life.mean(), np.median(life), np.percentile(life, 90)))
span = (t[:, 5].max() - t[:, 0].min()) * tick
print("first entry -> last exit: %.1f us" % span)
cu = t[:, 6]
ids = np.unique(cu)
per = np.array([(cu == i).sum() for i in ids])
print("%d distinct (xcc, se, cu) ids; workgroups per id min / mean / max = %d / %.1f / %d" % (len(ids), per.min(), per.mean(), per.max()))
# per CU: the fraction of its span during which >= 1 (and >= 2) workgroups were inside the tap loop
occ1, occ2, gaps = [], [], []
for i in ids[:: max(1, len(ids) // 32)]:
    rows = t[cu == i]
    ev = sorted([(r[2], 1) for r in rows] + [(r[3], -1) for r in rows])
    lo, hi = rows[:, 0].min(), rows[:, 5].max()
    cur, last, a1, a2 = 0, lo, 0, 0
    for tt, dlt in ev:
        if cur >= 1:
            a1 += tt - last
        if cur >= 2:
            a2 += tt - last
        cur += dlt
        last = tt
    occ1.append(a1 / max(1, hi - lo))
    occ2.append(a2 / max(1, hi - lo))
    starts = np.sort(rows[:, 0])
    ends = np.sort(rows[:, 5])
    gaps.append(np.median(starts[2:] - ends[:-2]) * tick if len(rows) > 4 else 0.0)      # exit of wg k -> entry of wg k + 2 (two slots per CU)
print("sampled CUs: >= 1 workgroup in its tap loop %.0f %% of the CU's span, >= 2: %.0f %%; slot turnaround (exit -> next entry) median %.2f us" % (
    100 * np.mean(occ1), 100 * np.mean(occ2), float(np.median(gaps))))
