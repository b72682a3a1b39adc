"""Where does the TRN head's time go?  Event-timed linears and wall-clock of the relation modules."""
import sys
import time
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, ".")
import pretorched_x_amd as ptx  # noqa: E402
L = ptx._lib
lib = L.lib()


def ev_time(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    t_ev = e0.elapsed_time(e1) / n
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return t_ev, (time.perf_counter() - t0) / n * 1e3


st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for (M, K, N) in [(8, 16384, 1024), (8, 4096, 1024), (8, 1024, 1024), (24, 14336, 1024), (64, 2048, 1000), (8, 2048, 339)]:
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    f = lambda: L.check(lib.ptx_linear_fwd(p(x), p(w), p(b), p(y), M, K, N, K, N, 9, st()))
    te, tw = ev_time(f)
    print("linear M=%-3d K=%-6d N=%-5d  %.1f us (wall %.1f us)  weights %.0f MB -> %.2f TB/s" % (
        M, K, N, te * 1e3, tw * 1e3, N * K * 4 / 1e6, N * K * 4 / te / 1e9))
x = torch.randn(8, 1, 8, 2048, device="cuda")
for name, mod in (("Relation(8,2048,1024,1024)", ptx.Relation(8, 2048, 1024, 1024)),
                  ("MultiScaleRelation(8,2048,1024,1024)", ptx.MultiScaleRelation(8, 2048, 1024, 1024))):
    mod = mod.cuda()
    np.random.seed(0)
    te, tw = ev_time(lambda: mod(x), 10)
    print("%-40s events %.3f ms  wall %.3f ms" % (name, te, tw))
