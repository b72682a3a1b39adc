"""Time ptx_maxpool3d_fwd on the I3D (config 4) pooling geometries; PTX_POOL_WSEG=4|8 picks the sliding segment.
   usage (GPU box): python scripts/gpu_pool_bench.py"""
import ctypes as C
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ptx = importlib.import_module("pretorched_x_amd")
L = ptx._lib
lib = L.lib()
SHAPES = [  # N, T, H, W, C, k, s  (TF-SAME, zero-valued padding)
    (2, 32, 112, 112, 64, (1, 3, 3), (1, 2, 2)), (2, 32, 56, 56, 192, (1, 3, 3), (1, 2, 2)),
    (2, 32, 28, 28, 192, (3, 3, 3), (1, 1, 1)), (2, 32, 28, 28, 256, (3, 3, 3), (1, 1, 1)),
    (2, 32, 28, 28, 480, (3, 3, 3), (2, 2, 2)), (2, 16, 14, 14, 480, (3, 3, 3), (1, 1, 1)),
    (2, 16, 14, 14, 512, (3, 3, 3), (1, 1, 1)), (2, 16, 14, 14, 528, (3, 3, 3), (1, 1, 1)),
    (2, 16, 14, 14, 832, (2, 2, 2), (2, 2, 2)), (2, 8, 7, 7, 832, (3, 3, 3), (1, 1, 1)),
    (8, 16, 14, 14, 512, (3, 3, 3), (1, 1, 1)), (8, 32, 28, 28, 192, (3, 3, 3), (1, 1, 1)),
    (8, 16, 112, 112, 64, (3, 3, 3), (2, 2, 2)), (8, 32, 56, 56, 64, (3, 3, 3), (2, 2, 2)),   # resnet3D.py:156 (configs 2, 3)
    (64, 1, 112, 112, 64, (1, 3, 3), (1, 2, 2))]                                              # 2-D resnet stem pool (TRN)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
tot_ms = 0.0
for (N, T, H, W, Cc, k, s) in SHAPES:
    out = [-(-i // q) for i, q in zip((T, H, W), s)]
    padt = [max((o - 1) * q + kk - i, 0) for o, q, kk, i in zip(out, s, k, (T, H, W))]
    fr = [t // 2 for t in padt]
    x = torch.randn(N, T, H, W, Cc, device="cuda")
    y = torch.empty(N, *out, Cc, device="cuda")
    d = L.PoolDesc(N, T, H, W, Cc, Cc, *out, *k, *s, *fr, Cc, L.PTX_POOL_SAME | L.PTX_POOL_PAD_ZERO)
    call = lambda: L.check(lib.ptx_maxpool3d_fwd(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), st), "pool")
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    mb = (x.numel() + y.numel()) * 4 / 1e6
    if N == 2:
        tot_ms += ms
    print(f"N={N} {T}x{H}x{W}x{Cc} k={k} s={s}: {ms*1e3:7.1f} us  {mb/ms/1e3:7.2f} TB/s")
# the ResNet3D stem pool exactly as the plans run it (MaxPool3d(3, 2, 1), resnet3D.py:156: symmetric padding, no SAME flags)
for (N, T, H, W, Cc) in [(8, 16, 112, 112, 64), (8, 32, 56, 56, 64)]:
    k, s_, p_ = (3, 3, 3), (2, 2, 2), (1, 1, 1)
    out = [(i + 2 - 3) // 2 + 1 for i in (T, H, W)]
    x = torch.randn(N, T, H, W, Cc, device="cuda")
    y = torch.empty(N, *out, Cc, device="cuda")
    d = L.PoolDesc(N, T, H, W, Cc, Cc, *out, *k, *s_, *p_, Cc, 0)
    call = lambda: L.check(lib.ptx_maxpool3d_fwd(C.byref(d), C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()), st), "pool")
    for _ in range(5):
        call()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 50
    mb = (x.numel() + y.numel()) * 4 / 1e6
    print(f"resnet stem pool N={N} {T}x{H}x{W}x{Cc}: {ms*1e3:7.1f} us  {mb/ms/1e3:7.2f} TB/s")
print(f"sum over the config-4 pools (N=2 rows, Mixed_4 x3 counted once each): {tot_ms:.3f} ms  WSEG={os.environ.get('PTX_POOL_WSEG', 'default')}")
