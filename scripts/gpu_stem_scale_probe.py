"""Where does the fp32 direct stem lose its last 25 %?  Time ptx_conv_stem_f32_fwd on config-2 geometry at several batch
sizes (tail / wave quantisation shows as a non-linear time) and on a one-round launch (pure per-workgroup time).
   usage (GPU box): python scripts/gpu_stem_scale_probe.py"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ptx = importlib.import_module("pretorched_x_amd")
L = ptx._lib
lib = L.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())


def run(N, T, H, W, kT, pT, label):
    d = L.ConvDesc()
    Ho, Wo = H // 2, W // 2
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, 3, 0
    d.To, d.Ho, d.Wo, d.Co, d.ldy = T, Ho, Wo, 64, 64
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = kT, 7, 7, 1, 2, 2, pT, 3, 3
    d.Kc, d.Co_pad, d.flags = 24, 128, L.PTX_EPI_RELU
    plane = H * W
    sn, sc, st = 3 * T * plane, T * plane, plane
    assert lib.ptx_conv_stem_f32_supported(C.byref(d), sn, sc, st)
    x = torch.randn(N, 3, T, H, W, device="cuda")
    w = torch.randn(lib.ptx_stem_f32_weight_elems(C.byref(d)), device="cuda") * 0.03
    y = torch.empty(N, T, Ho, Wo, 64, device="cuda")
    bias = torch.zeros(128, device="cuda")
    call = lambda: L.check(lib.ptx_conv_stem_f32_fwd(C.byref(d), p(x), sn, sc, st, p(w), p(bias), p(y), s), "stem")
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    taps = sum(min(kT - 1, T - 1 - (to - pT)) - max(0, pT - to) + 1 for to in range(T))
    flop = 2.0 * N * taps * 49 * 3 * Ho * Wo * 64           # issued (pruned) work
    wgs = N * T * -(-Ho * Wo // 512)
    steps = N * taps * 7 * -(-Ho * Wo // 512)
    print(f"{label:28s} N={N:3d} T={T:2d} {H}x{W}: {ms:8.4f} ms  {flop / ms / 1e9:6.1f} TF issued  {wgs:5d} WGs = {wgs / 256:6.2f} per CU"
          f"  {ms * 1e3 / (steps / 256):7.3f} us per (kt,kh) step per CU-slot")


if len(sys.argv) > 1:            # one batch size only (under rocprofv3)
    run(int(sys.argv[1]), 16, 224, 224, 7, 3, "config-2 stem")
    sys.exit(0)
for N in (2, 4, 8, 16, 32):
    run(N, 16, 224, 224, 7, 3, "config-2 stem")
run(10, 1, 224, 224, 1, 0, "2-D stem, 250 WGs: one round")
run(20, 1, 224, 224, 1, 0, "2-D stem, 500 WGs")
run(2, 5, 224, 224, 7, 3, "T=5: 250 WGs, 4-5 taps")
run(8, 16, 112, 112, 7, 3, "config-3-like 112x112")
