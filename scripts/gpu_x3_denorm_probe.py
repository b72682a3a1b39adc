"""Do the fp16 MFMAs keep subnormal half inputs?  Relative error of an x3 conv when the operands are scaled down so that
the lo halves are subnormal (|v| < 0.125 -> lo < 2^-14): ~1e-6 if subnormals are honoured, ~1e-4 if flushed."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import pretorched_x_amd as ptx
import test_gpu_kernels as T
lib = ptx._lib.lib()
names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
for scale_x, scale_w in ((1.0, 0.05), (1e-2, 0.05), (1e-3, 1e-3), (1e-4, 1.0)):
    x = T.rnd(2, 64, 3, 9, 10, seed=4) * scale_x
    w = T.rnd(96, 64, 3, 3, 3, seed=5) * scale_w
    want = T.ref_conv(x.double(), w.double(), (1, 1, 1), (1, 1, 1)).float()
    out = []
    for tag in ("128x128x32/4x2/m32/dma/x3", "32x64x64/2x2/m16/dma/x3"):
        got = T.hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), cfg=names.index(tag), split=1, x3=True)
        out.append("%s rel %.2e" % (tag.split("/")[2], (got - want).abs().max().item() / want.abs().max().item()))
    got32 = T.hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1))
    print("x~%g w~%g max|y| %.2e:" % (scale_x, scale_w, want.abs().max().item()), " ".join(out),
          "fp32 rel %.2e" % ((got32 - want).abs().max().item() / want.abs().max().item()))
