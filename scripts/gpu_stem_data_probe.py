"""Is the fp32 stem power/clock bound?  Time ptx_conv_stem_f32_fwd (config-2 geometry) on zero, constant and random
inputs / filters: identical instruction streams, different operand toggling.
   usage (GPU box): python scripts/gpu_stem_data_probe.py"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ptx = importlib.import_module("pretorched_x_amd")
L = ptx._lib
lib = L.lib()
N, T, H, W, Co = 8, 16, 224, 224, 64
d = L.ConvDesc()
d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, 3, 0
d.To, d.Ho, d.Wo, d.Co, d.ldy = 16, 112, 112, Co, 64
d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 7, 7, 7, 1, 2, 2, 3, 3, 3
d.Kc, d.Co_pad, d.flags = 24, 128, L.PTX_EPI_RELU
plane = H * W
sn, sc, st = 3 * T * plane, T * plane, plane
assert lib.ptx_conv_stem_f32_supported(C.byref(d), sn, sc, st)
nw = lib.ptx_stem_f32_weight_elems(C.byref(d))
y = torch.empty(N, 16, 112, 112, 64, device="cuda")
bias = torch.zeros(128, device="cuda")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
flop = 2.0 * N * 16 * 112 * 112 * 64 * 3 * 343
for xname, wname in [("zeros", "zeros"), ("ones", "ones"), ("randn", "ones"), ("ones", "randn"), ("randn", "randn"), ("zeros", "zeros")]:
    mk = {"zeros": torch.zeros, "ones": torch.ones, "randn": torch.randn}
    x = mk[xname](N, 3, T, H, W, device="cuda")
    w = mk[wname](nw, device="cuda") * 0.03
    call = lambda: L.check(lib.ptx_conv_stem_f32_fwd(C.byref(d), p(x), sn, sc, st, p(w), p(bias), p(y), s), "stem")
    for _ in range(3):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"x={xname:6s} w={wname:6s}: {ms:.4f} ms  {flop / ms / 1e9:7.1f} TF algorithmic")
