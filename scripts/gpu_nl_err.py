"""Error of ptx_nonlocal_fwd vs torch fp64 per mode (fp32 / x3 / f16 MFMAs) -- a numerics probe, not a test."""
import ctypes as C, sys, os
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pretorched_x_amd as ptx
L = ptx._lib; lib = L.lib()
def p(t, off=0): return C.c_void_p(t.data_ptr() + 4 * off)
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
for (B, Nq, Nk, d, dv, mode) in [(2, 196, 196, 512, 512, "softmax"), (1, 300, 300, 256, 256, "softmax"), (2, 90, 90, 40, 24, "scale"),
                                 (2, 256, 64, 64, 256, "softmax"), (2, 90, 90, 40, 24, "softmax")]:
    g = torch.Generator().manual_seed(1000 + Nq + d)
    ld = (2 * d + dv + 3) // 4 * 4 + 4
    tq = torch.randn(B, Nq, ld, generator=g); tk = torch.randn(B, Nk, ld, generator=g)
    tq[..., :d] *= 3.0 / d ** 0.5
    theta, phi, gv = tq[..., :d].double(), tk[..., d:2 * d].double(), tk[..., 2 * d:2 * d + dv].double()
    f = theta @ phi.transpose(1, 2)
    f = F.softmax(f, -1) if mode == "softmax" else f / f.size(-1)
    want = (f @ gv)
    tqd, tkd = tq.cuda(), tk.cuda()
    ldy = (dv + 3) // 4 * 4 + 8
    out = []
    for name, bits in (("fp32", 0), ("x3", L.PTX_NL_X3), ("f16", L.PTX_NL_F16)):
        if bits == L.PTX_NL_F16 and (mode != "softmax" or d > 64): continue
        y = torch.zeros(B, Nq, ldy, device="cuda")
        desc = L.NonlocalDesc()
        desc.batch, desc.Nq, desc.Nk, desc.d, desc.dv = B, Nq, Nk, d, dv
        desc.ld_theta = desc.ld_phi = desc.ld_g = ld; desc.ld_y = ldy
        desc.bs_theta, desc.bs_phi, desc.bs_g, desc.bs_y = Nq * ld, Nk * ld, Nk * ld, Nq * ldy
        desc.mode = (L.PTX_NL_SOFTMAX if mode == "softmax" else L.PTX_NL_SCALE) | bits
        L.check(lib.ptx_nonlocal_fwd(C.byref(desc), p(tqd), p(tkd, d), p(tkd, 2 * d), p(y), st()), name)
        torch.cuda.synchronize()
        out.append("%s %.2e" % (name, (y.cpu()[..., :dv].double() - want).abs().max().item()))
    print((B, Nq, Nk, d, dv, mode), "max|want| %.2f" % want.abs().max().item(), " ".join(out))
