"""Round-6 probe: ptx_conv_body_f32_fwd (patch-resident 3x3x3 kernel) against the generic implicit-GEMM tiles on the 3x3x3
problems of config 2, same box, same process.  usage (GPU box): python scripts/gpu_body_probe.py [clips]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pretorched_x_amd as ptx  # noqa: E402

L, lib = ptx._lib, ptx._lib.lib()
DEV = "cuda:0"
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 8
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)      # noqa: E731
p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731


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


# (the 64 x 64 x 6-frame case has NO tail tiles: 768 tall workgroups = exactly three per CU -- the kernel's loop without the
#  tile-quantisation loss of the 56 x 56 frames)
for name, C_, T, H in (("layer1.conv2 (C3)", 64, 8, 56), ("balanced 64x64x6", 64, 6, 64), ("layer2.conv2 (C11)", 128, 4, 28), ("layer3.conv2 (C17)", 256, 2, 14)):
    N, Co, W = clips, C_, H
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, T, H, W, C_, generator=g).to(DEV)
    w = (torch.randn(Co, C_, 3, 3, 3, generator=g) * (2.0 / (C_ * 27)) ** 0.5).to(DEV)
    pd = L.PackDesc(Co, C_, 3, 3, 3, C_, (Co + 127) // 128 * 128, 0)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(pd.Co_pad, device=DEV)
    null = C.c_void_p(0)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), p(w), null, null, null, null, null, C.c_float(0.0), p(wp), p(bp), st()), "pack")
    y0 = torch.empty(N, T, H, W, Co, device=DEV)
    y1 = torch.empty_like(y0)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, C_, C_
    d.To, d.Ho, d.Wo, d.Co, d.ldy = T, H, W, Co, Co
    d.kT = d.kH = d.kW = 3
    d.sT = d.sH = d.sW = 1
    d.pT = d.pH = d.pW = 1
    d.Kc, d.Co_pad, d.flags = pd.Kc, pd.Co_pad, L.PTX_EPI_RELU
    macs = N * T * H * W * Co * C_ * 27
    ws_bytes = lib.ptx_conv3d_workspace_bytes(C.byref(d), 8)
    ws = torch.empty(max(ws_bytes // 4, 4), device=DEV)
    from pretorched_x_amd import engine as E
    import json
    ent = E.tuned_lookup(json.dumps(d.key()), "")
    cfg, split = ent if ent is not None else (-1, 0)
    t_gen = timed(lambda: L.check(lib.ptx_conv3d_fwd(C.byref(d), p(x), p(wp), p(bp), null, p(y0), p(ws), ws_bytes, cfg, split, st()), "conv"))
    gname = lib.ptx_conv3d_config_name(cfg).decode() if cfg >= 0 else "default"
    print("%-20s clips=%d  generic %-28s split=%d  %.4f ms  %.1f TF" % (name, N, gname, split, t_gen, 2e-9 * macs / t_gen), flush=True)
    for shape in (0, 1):
        if not lib.ptx_conv_body_f32_supported(C.byref(d), shape):
            print("%-20s shape %d: not supported" % (name, shape))
            continue
        wb = torch.empty(lib.ptx_conv_body_f32_weight_elems(C.byref(d)), device=DEV)
        L.check(lib.ptx_pack_conv_body_f32_weight(C.byref(d), p(wp), p(wb), st()), "pack body")
        y1.fill_(float("nan"))
        t_b = timed(lambda: L.check(lib.ptx_conv_body_f32_fwd(C.byref(d), p(x), p(wb), p(bp), null, p(y1), shape, st()), "body"))
        err = (y1 - y0).abs().max().item()
        print("%-20s clips=%d  body shape %d %37s %.4f ms  %.1f TF   max|d vs generic| = %.2e (max|y| %.2f)" % (
            name, N, shape, "", t_b, 2e-9 * macs / t_b, err, y0.abs().max().item()), flush=True)


# ---- the chained tail: layer1.{1,2}.conv2 + conv3 (+ residual + ReLU) of config 2, body kernel vs ptx_conv3d_chain_fwd ----
N, C_, N1, N2, T, H, W = clips, 64, 64, 256, 8, 56, 56
g = torch.Generator().manual_seed(2)
x = torch.randn(N, T, H, W, C_, generator=g).to(DEV)
res = torch.randn(N, T, H, W, N2, generator=g).to(DEV)
w1 = (torch.randn(N1, C_, 3, 3, 3, generator=g) * (2.0 / (C_ * 27)) ** 0.5).to(DEV)
w2 = (torch.randn(N2, N1, 1, 1, 1, generator=g) * (2.0 / N1) ** 0.5).to(DEV)
null = C.c_void_p(0)


def pack(w):
    Co, Ci, kT, kH, kW = w.shape
    pd = L.PackDesc(Co, Ci, kT, kH, kW, Ci, (Co + 127) // 128 * 128, 0)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(pd.Co_pad, device=DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), p(w), null, null, null, null, null, C.c_float(0.0), p(wp), p(bp), st()), "pack")
    return pd, wp, bp


pd1, wp1, bp1 = pack(w1)
pd2, wp2, bp2 = pack(w2)
d = L.ConvDesc()
d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, C_, C_
d.To, d.Ho, d.Wo, d.Co, d.ldy = T, H, W, N1, N1
d.kT = d.kH = d.kW = 3
d.sT = d.sH = d.sW = 1
d.pT = d.pH = d.pW = 1
d.Kc, d.Co_pad, d.flags, d.groups = pd1.Kc, pd1.Co_pad, L.PTX_EPI_RELU, 1
t = L.ConvDesc()
t.N, t.Ti, t.Hi, t.Wi, t.Ci, t.ldx = N, T, H, W, N1, N1
t.To, t.Ho, t.Wo, t.Co, t.ldy = T, H, W, N2, N2
t.kT = t.kH = t.kW = t.sT = t.sH = t.sW = 1
t.Kc, t.Co_pad, t.flags, t.ldr, t.groups = pd2.Kc, pd2.Co_pad, L.PTX_EPI_RELU | L.PTX_EPI_RES_ADD, N2, 1
y0 = torch.empty(N, T, H, W, N2, device=DEV)
y1 = torch.empty_like(y0)
macs = N * T * H * W * (N1 * C_ * 27 + N2 * N1)
best = None
for cfg in range(lib.ptx_conv3d_chain_num_configs()):
    if not lib.ptx_conv3d_chain_supported(C.byref(d), C.byref(t), cfg):
        continue
    ms = timed(lambda: L.check(lib.ptx_conv3d_chain_fwd(C.byref(d), C.byref(t), p(x), p(wp1), p(bp1), p(wp2), p(bp2), p(res), p(y0), cfg, st()), "chain"), 10)
    if best is None or ms < best[0]:
        best = (ms, cfg)
L.check(lib.ptx_conv3d_chain_fwd(C.byref(d), C.byref(t), p(x), p(wp1), p(bp1), p(wp2), p(bp2), p(res), p(y0), best[1], st()), "chain")
print("layer1 conv2+conv3     clips=%d  chained tile %-34s %.4f ms  %.1f TF" % (N, lib.ptx_conv3d_chain_config_name(best[1]).decode(), best[0], 2e-9 * macs / best[0]), flush=True)
wb = torch.empty(lib.ptx_conv_body_f32_weight_elems(C.byref(d)), device=DEV)
wt = torch.empty(lib.ptx_conv_body_tail_f32_weight_elems(C.byref(t)), device=DEV)
L.check(lib.ptx_pack_conv_body_f32_weight(C.byref(d), p(wp1), p(wb), st()), "pack body")
L.check(lib.ptx_pack_conv_body_tail_f32_weight(C.byref(t), p(wp2), p(wt), st()), "pack tail")
for shape in (0, 1):
    if not lib.ptx_conv_body_chain_f32_supported(C.byref(d), C.byref(t), shape):
        continue
    y1.fill_(float("nan"))
    ms = timed(lambda: L.check(lib.ptx_conv_body_chain_f32_fwd(C.byref(d), C.byref(t), p(x), p(wb), p(bp1), p(wt), p(bp2), p(res), p(y1), shape, st()), "body chain"))
    print("layer1 conv2+conv3     clips=%d  body chain shape %d %28s %.4f ms  %.1f TF   max|d vs chained tile| = %.2e (max|y| %.2f)" % (
        N, shape, "", ms, 2e-9 * macs / ms, (y1 - y0).abs().max().item(), y0.abs().max().item()), flush=True)
