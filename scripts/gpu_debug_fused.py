"""debug helper: run the fused-stage kernel test cases per tile configuration and report which fail"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import pretorched_x_amd as ptx
import test_gpu_kernels as T
lib = ptx._lib.lib()
names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
f16 = [i for i, n in enumerate(names) if n.endswith("/f16")]
cases = {
 "up2 3x3 Ci32 7x5": (2, 7, 5, 32, 32, 3, True, True, True, False, True, False, None),
 "up2 3x3 Ci64 8x8": (2, 8, 8, 64, 32, 3, True, True, True, False, True, False, None),
 "up2 3x3 Ci64 7x5": (2, 7, 5, 64, 32, 3, True, True, True, False, True, False, None),
 "up2 3x3 Ci32 8x8": (2, 8, 8, 32, 32, 3, True, True, True, False, True, False, None),
 "noup 3x3 Ci32 14x10": (2, 14, 10, 32, 32, 3, False, True, True, False, True, False, None),
 "up2 1x1 Ci64 7x5": (2, 7, 5, 64, 32, 1, True, True, True, False, True, False, None),
 "up2 3x3 plain fp32 out": (2, 7, 5, 64, 32, 3, True, False, False, False, False, False, None),
}
for cname, args in cases.items():
    for cfg in [-1] + f16:
        for split in ((0,) if cfg < 0 else (1, 2)):
            try:
                T._fused_stage_case(ptx, *args, [(cfg, split)])
                r = "ok"
            except AssertionError as e:
                r = "FAIL " + str(e)[:60]
            except Exception as e:
                r = "ERR " + str(e)[:80]
            if r != "ok":
                print(cname, names[cfg] if cfg >= 0 else "auto", "split", split, r)
    print(cname, "done")
