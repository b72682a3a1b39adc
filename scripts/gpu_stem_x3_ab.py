"""Same-box A/B of the split-operand stem: the shipped kernel (64-row wave tiles, one 114-KB workgroup per CU) against the
round-3 experiment with 32-row wave tiles and two 80-KB workgroups per CU (scripts/micro/conv_stem_x3_2wg.hip, built by
hand into scripts/micro/libptx_stem2wg.so -- see its header; not part of the product).  Config-2 / I3D / (2+1)D geometries.
   usage (GPU box): python scripts/gpu_stem_x3_ab.py"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import ctypes as C, importlib, os, sys, torch
sys.path.insert(0, %r)
ptx = importlib.import_module("pretorched_x_amd")
L = ptx._lib
if os.environ.get("PTX_AB_LIB"):
    L.LIB_PATH = os.environ["PTX_AB_LIB"]
lib = L.lib()
def run(N, T, H, W, k, s_, p_, To, Ho, Wo, tag):
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, 3, 4
    d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, 64, 64
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = k[0], k[1], k[2], s_[0], s_[1], s_[2], p_[0], p_[1], p_[2]
    d.Kc, d.Co_pad, d.flags = 32, 128, L.PTX_EPI_RELU | L.PTX_F16X3_OPERANDS
    assert lib.ptx_conv_stem_x3_supported(C.byref(d))
    pd = L.PackDesc(64, 4, k[0], k[1], k[2], 32, 128, 1)
    pd.f16 = 2
    nw = lib.ptx_packed_weight_elems(C.byref(pd))
    x = (torch.randn(2 * N * T * H * W * 4, device="cuda") * 0.5).to(torch.float16).view(torch.float32)
    w = (torch.randn(2 * nw, device="cuda") * 0.05).to(torch.float16).view(torch.float32)
    y = torch.empty(N, To, Ho, Wo, 64, device="cuda")
    b = torch.zeros(128, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    call = lambda: L.check(lib.ptx_conv_stem_x3_fwd(C.byref(d), p(x), p(w), p(b), p(y), st), "stem")
    best = 1e9
    for rep in range(3):
        for _ in range(3): call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): call()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20)
    print("%%-28s %%-34s %%.4f ms" %% (tag, os.path.basename(L.LIB_PATH), best), flush=True)
run(8, 16, 224, 224, (7, 7, 7), (1, 2, 2), (3, 3, 3), 16, 112, 112, "resnet3d50 stem (config 2)")
run(2, 64, 224, 224, (7, 7, 7), (2, 2, 2), (3, 3, 3), 32, 112, 112, "I3D stem (config 4 share)")
run(8, 32, 112, 112, (1, 7, 7), (1, 2, 2), (0, 3, 3), 32, 56, 56, "(2+1)D spatial stem (cfg 3)")
''' % ROOT
old = os.path.join(ROOT, "scripts", "micro", "libptx_stem2wg.so")
for rep in range(2):
    for lib in ("", old):
        env = dict(os.environ, PTX_AB_LIB=lib)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
