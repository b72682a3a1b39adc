"""Micro-benchmark ptx_conv3d_fwd on synthetic geometries: every (config, split) requested.

    python scripts/conv_micro.py "N,T,H,W,Ci,Co,k,s,p[,res]" ... [--cfgs 0,1,2] [--splits 1,2]
"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
import pretorched_x_amd as ptx  # noqa: E402

L = ptx._lib
lib = L.lib()
DEV = "cuda:0"


def p(t):
    return C.c_void_p(t.data_ptr())


def st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(spec, cfgs, splits, iters=10):
    f = spec.split(",")
    N, T, H, W, Ci, Co, k, s, pad = [int(v) for v in f[:9]]
    res = len(f) > 9 and f[9] == "res"
    kT = kH = kW = k
    if k > 100:        # e.g. 133 -> (1,3,3); 311 -> (3,1,1)
        kT, kH, kW = k // 100, (k // 10) % 10, k % 10
    pT, pH, pW = (kT // 2, kH // 2, kW // 2) if pad else (0, 0, 0)
    To, Ho, Wo = (T + 2 * pT - kT) // s + 1, (H + 2 * pH - kH) // s + 1, (W + 2 * pW - kW) // s + 1
    Kc, Co_pad = (Ci + 3) // 4 * 4, (Co + 127) // 128 * 128
    x = torch.randn(N, T, H, W, Kc, device=DEV)
    w = torch.randn(kT * kH * kW * Co_pad * Kc, device=DEV) * 0.05
    b = torch.randn(Co_pad, device=DEV)
    y = torch.empty(N, To, Ho, Wo, Co, device=DEV)
    r = torch.randn(N, To, Ho, Wo, Co, device=DEV) if res else None
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, Kc
    d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, Co, Co
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = kT, kH, kW, s, s, s, pT, pH, pW
    d.Kc, d.Co_pad = Kc, Co_pad
    d.flags = L.PTX_EPI_RELU | (L.PTX_EPI_RES_ADD if res else 0)
    d.ldr = Co
    M = N * To * Ho * Wo
    flop = 2.0 * M * Co * Ci * kT * kH * kW
    ws_bytes = lib.ptx_conv3d_workspace_bytes(C.byref(d), max(splits))
    ws = torch.empty(max(ws_bytes // 4, 4), device=DEV)
    print("## %s  M=%d N=%d K=%d  %.2f GFLOP" % (spec, M, Co, Ci * kT * kH * kW, flop / 1e9))
    for cfg in cfgs:
        for sk in splits:
            args = (C.byref(d), p(x), p(w), p(b), p(r) if res else None, p(y), p(ws), ws_bytes, cfg, sk, st())
            if lib.ptx_conv3d_fwd(*args) != 0:
                print("   cfg %d split %d: %s" % (cfg, sk, lib.ptx_last_error().decode()))
                continue
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                lib.ptx_conv3d_fwd(*args)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / iters
            print("   %-22s split=%d  %8.4f ms  %6.1f TF  (%4.1f%%)" % (lib.ptx_conv3d_config_name(cfg).decode(), sk, ms,
                                                                     flop / ms / 1e9, flop / ms / 1e9 / 1.573))


if __name__ == "__main__":
    specs = [a for a in sys.argv[1:] if not a.startswith("--")]
    cfgs = list(range(lib.ptx_conv3d_num_configs()))
    splits = [1]
    for a in sys.argv[1:]:
        if a.startswith("--cfgs="):
            cfgs = [int(v) for v in a[7:].split(",")]
        if a.startswith("--splits="):
            splits = [int(v) for v in a[9:].split(",")]
    for s in specs:
        run(s, cfgs, splits)
