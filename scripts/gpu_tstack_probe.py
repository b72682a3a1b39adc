"""Round-6 probe: the T-stacked tile of ptx_conv_body_f32_fwd ((kT,1,1) temporal convs) against the generic implicit-GEMM
tiles on config 3's temporal problems, same box, same process.  usage (GPU box): python scripts/gpu_tstack_probe.py [clips]"""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd import engine as E  # noqa: E402

L, lib = ptx._lib, ptx._lib.lib()
DEV = "cuda:0"
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)      # noqa: E731
p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
null = C.c_void_p(0)


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / iters


CASES = (("conv1.temporal (7,1,1) 110->64", 110, 64, 7, 32, 56),
         ("layer1 conv2.temporal 144->64", 144, 64, 3, 16, 28),
         ("layer2 conv2.temporal 288->128", 288, 128, 3, 8, 14),
         ("layer3 conv2.temporal 576->256", 576, 256, 3, 4, 7))
for name, Ci, Co, kT, T, H in CASES:
    N, W = clips, H
    Kc = (Ci + 3) // 4 * 4
    g = torch.Generator().manual_seed(1)
    x = torch.zeros(N, T, H, W, Kc)
    x[..., :Ci] = torch.randn(N, T, H, W, Ci, generator=g)
    x = x.to(DEV)
    w = (torch.randn(Co, Ci, kT, 1, 1, generator=g) * (2.0 / (Ci * kT)) ** 0.5).to(DEV)
    pd = L.PackDesc(Co, Ci, kT, 1, 1, Kc, (Co + 127) // 128 * 128, 0)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(pd.Co_pad, device=DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), p(w), null, null, null, null, null, C.c_float(0.0), p(wp), p(bp), st()), "pack")
    y0 = torch.empty(N, T, H, W, Co, device=DEV)
    y1 = torch.empty_like(y0)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, Kc
    d.To, d.Ho, d.Wo, d.Co, d.ldy = T, H, W, Co, Co
    d.kT, d.kH, d.kW = kT, 1, 1
    d.sT = d.sH = d.sW = 1
    d.pT, d.pH, d.pW = kT // 2, 0, 0
    d.Kc, d.Co_pad, d.flags = pd.Kc, pd.Co_pad, L.PTX_EPI_RELU
    macs = N * T * H * W * Co * Ci * kT
    ws_bytes = lib.ptx_conv3d_workspace_bytes(C.byref(d), 8)
    ws = torch.empty(max(ws_bytes // 4, 4), device=DEV)
    ent = E.tuned_lookup(json.dumps(d.key()), "")
    rows = []
    for cfg in range(lib.ptx_conv3d_num_configs()):
        cname = lib.ptx_conv3d_config_name(cfg).decode()
        if any(s in cname for s in ("f16", "x3", "direct", "kwr", "chain")) or not lib.ptx_conv3d_config_supported(C.byref(d), cfg):
            continue
        for split in ((1,) if macs > 2e9 else (1, 2, 4)):
            try:
                ms = timed(lambda: L.check(lib.ptx_conv3d_fwd(C.byref(d), p(x), p(wp), p(bp), null, p(y0), p(ws), ws_bytes, cfg, split, st()), "conv"), 8)
            except Exception:
                continue
            rows.append((ms, cname, split, cfg))
    rows.sort()
    print("== %s  clips=%d  M=%d  %.2f GFLOP   tuned table: %s" % (name, N, N * T * H * W, 2e-9 * macs, ent), flush=True)
    for ms, cname, split, cfg in rows[:3]:
        print("   generic %-32s split=%d  %.4f ms  %.1f TF" % (cname, split, ms, 2e-9 * macs / ms))
    ms, cname, split, cfg = rows[0]
    L.check(lib.ptx_conv3d_fwd(C.byref(d), p(x), p(wp), p(bp), null, p(y0), p(ws), ws_bytes, cfg, split, st()), "conv")
    if not lib.ptx_conv_body_f32_supported(C.byref(d), 0):
        print("   T-stacked tile: not supported")
        continue
    wb = torch.empty(lib.ptx_conv_body_f32_weight_elems(C.byref(d)), device=DEV)
    L.check(lib.ptx_pack_conv_body_f32_weight(C.byref(d), p(wp), p(wb), st()), "pack body")
    y1.fill_(float("nan"))
    t_b = timed(lambda: L.check(lib.ptx_conv_body_f32_fwd(C.byref(d), p(x), p(wb), p(bp), null, p(y1), 0, st()), "tstack"))
    err = (y1 - y0).abs().max().item()
    print("   T-stacked tile %34s %.4f ms  %.1f TF   max|d vs generic| = %.2e (max|y| %.2f)" % (
        "", t_b, 2e-9 * macs / t_b, err, y0.abs().max().item()), flush=True)
