"""Time the HBM-bound side kernels at config-2 shapes (maxpool, fold_kw)."""
import ctypes as C, sys, torch
sys.path.insert(0, ".")
import pretorched_x_amd as ptx
L = ptx._lib; lib = L.lib()
p = lambda t: C.c_void_p(t.data_ptr())
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
x = torch.randn(8, 16, 112, 112, 64, device="cuda"); y = torch.empty(8, 8, 56, 56, 64, device="cuda")
d = L.PoolDesc(8, 16, 112, 112, 64, 64, 8, 56, 56, 3, 3, 3, 2, 2, 2, 1, 1, 1)
ms = timeit(lambda: L.check(lib.ptx_maxpool3d_fwd(C.byref(d), p(x), p(y), st())))
ref = torch.nn.functional.max_pool3d(x.permute(0, 4, 1, 2, 3), 3, 2, 1).permute(0, 2, 3, 4, 1)
print("maxpool3d cfg2: %.1f us  %.2f TB/s  exact=%s" % (ms * 1e3, (x.numel() + y.numel()) * 4 / ms / 1e9, torch.equal(ref, y)))
xi = torch.randn(8, 3, 16, 224, 224, device="cuda"); xo = torch.empty(8, 16, 224, 112, 24, device="cuda")
ms = timeit(lambda: L.check(lib.ptx_fold_kw_ncdhw(p(xi), p(xo), 8, 3, 16, 224, 224, 7, 2, 3, 112, 24, st())))
print("fold_kw cfg2: %.1f us  %.2f TB/s" % (ms * 1e3, (xi.numel() + xo.numel()) * 4 / ms / 1e9))
