"""Is the N = 1568 attention kernel bound per CU or chip-wide?  One launch at 2 .. 10 clips (25 workgroups of 8 waves per clip,
one per CU up to 10 clips) with and without the stream-K workspace.  Measurement only."""
import ctypes as C
import os
import sys

import torch

os.environ["PTX_NL_STREAMK"] = "1"        # the form is off by default

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pretorched_x_amd as ptx  # noqa: E402

L = ptx._lib
lib = L.lib()
N, d, dv = 1568, 256, 256
ld = 768
for B in (2, 4, 6, 8, 10, 16):
    tpg = torch.randn(B, N, ld, device="cuda") * 0.2
    y = torch.empty(B, N, dv, device="cuda")
    desc = L.NonlocalDesc()
    desc.batch, desc.Nq, desc.Nk, desc.d, desc.dv = B, N, N, d, dv
    desc.ld_theta = desc.ld_phi = desc.ld_g = ld
    desc.ld_y = dv
    desc.bs_theta = desc.bs_phi = desc.bs_g = N * ld
    desc.bs_y = N * dv
    desc.mode = L.PTX_NL_SOFTMAX
    need = lib.ptx_nonlocal_workspace_bytes(C.byref(desc))
    ws = torch.empty(max(need // 4, 4), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t, off=0: C.c_void_p(t.data_ptr() + 4 * off)   # noqa: E731
    out = []
    for wsp, nb in ((None, 0), (p(ws), need)):
        for _ in range(5):
            L.check(lib.ptx_nonlocal_ws_fwd(C.byref(desc), p(tpg), p(tpg, d), p(tpg, 2 * d), p(y), wsp, nb, st), "nl")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            L.check(lib.ptx_nonlocal_ws_fwd(C.byref(desc), p(tpg), p(tpg, d), p(tpg, 2 * d), p(y), wsp, nb, st), "nl")
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / 30 * 1e3)
    flop = 2.0 * B * N * N * (d + dv)
    print("clips %2d  plain (%3d workgroups) %7.1f us %6.1f TF | stream-K (%3d workgroups) %7.1f us %6.1f TF" % (
        B, 25 * B, out[0], flop / out[0] / 1e6, 32 * B, out[1], flop / out[1] / 1e6), flush=True)
