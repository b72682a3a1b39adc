"""Round-6 probe: do the small-M launches (about one 4-wave workgroup per CU) leave the CU idle in a way a SECOND resident
workgroup would fill?  The same launch is issued once, and twice concurrently on two HIP streams (two independent output buffers):
if the pair takes about as long as one, a CU runs two such workgroups at nearly twice the rate -- the case for 8-wave workgroups
that split K inside the workgroup.    python scripts/gpu_r06_coresident_probe.py"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import pretorched_x_amd as ptx  # noqa: E402

L, lib = ptx._lib, ptx._lib.lib()
DEV = "cuda:0"
p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
null = C.c_void_p(0)


def setup(N, T, H, W, Ci, Co, kT, kH, kW, cfg_name):
    Kc, Co_pad = (Ci + 3) // 4 * 4, (Co + 127) // 128 * 128
    x = torch.randn(N, T, H, W, Kc, device=DEV)
    w = torch.randn(kT * kH * kW * Co_pad * Kc, device=DEV) * 0.05
    b = torch.randn(Co_pad, device=DEV)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, Kc
    d.To, d.Ho, d.Wo, d.Co, d.ldy = T, H, W, Co, (Co + 3) // 4 * 4
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = kT, kH, kW, 1, 1, 1, kT // 2, kH // 2, kW // 2
    d.Kc, d.Co_pad, d.flags = Kc, Co_pad, L.PTX_EPI_RELU
    cfg = next(i for i in range(lib.ptx_conv3d_num_configs()) if lib.ptx_conv3d_config_name(i).decode() == cfg_name)
    ys = [torch.empty(N, T, H, W, d.ldy, device=DEV) for _ in range(2)]
    ws = [torch.empty(1 << 20, device=DEV) for _ in range(2)]
    return d, x, w, b, ys, ws, cfg


def run(name, *args):
    d, x, w, b, ys, ws, cfg = setup(*args)
    s0, s1 = torch.cuda.current_stream(), torch.cuda.Stream()
    st = lambda s: C.c_void_p(s.cuda_stream)      # noqa: E731

    def one(i, s):
        L.check(lib.ptx_conv3d_fwd(C.byref(d), p(x), p(w), p(b), null, p(ys[i]), p(ws[i]), 4 << 20, cfg, 1, st(s)), "conv")

    def timed(pair, iters=50):
        for _ in range(5):
            one(0, s0)
            if pair:
                one(1, s1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s1.wait_stream(s0)
        e0.record(s0)
        s1.wait_event(e0)
        for _ in range(iters):
            one(0, s0)
            if pair:
                one(1, s1)
        s0.wait_stream(s1)
        e1.record(s0)
        e1.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    a, b2 = timed(False), timed(True)
    M = d.N * d.To * d.Ho * d.Wo
    print("%-44s M=%-5d one launch %6.2f us | two concurrent launches %6.2f us per PAIR (%.2fx one)" % (name, M, a, b2, b2 / a), flush=True)


run("cfg3 layer3 conv2.temporal 576->256 (3,1,1)", 8, 4, 7, 7, 576, 256, 3, 1, 1, "32x64x64/2x2/m16/dma")
run("cfg3 layer3 conv1.spatial 1024->204", 8, 4, 7, 7, 1024, 204, 1, 1, 1, "32x64x64/2x2/m16/dma/re")
run("cfg3 layer3 conv2.spatial 256->576 (1,3,3)", 8, 4, 7, 7, 256, 576, 1, 3, 3, "32x64x64/2x2/m16/dma")
run("cfg3 layer4 conv1.temporal 409->512 (M=256)", 8, 2, 4, 4, 412, 512, 1, 1, 1, "32x64x64/2x2/m16/dma/re")
run("cfg2 layer3 conv1 1024->256 (M=3136)", 8, 2, 14, 14, 1024, 256, 1, 1, 1, "32x64x64/2x2/m16/dma/re")
run("cfg2 layer2 conv1 512->128 (M=25088)", 8, 4, 28, 28, 512, 128, 1, 1, 1, "32x64x32/2x2/m16/dma/re")
