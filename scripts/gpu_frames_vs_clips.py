"""VERDICT r2 #5 'Done' check: model.forward_frames (uint8 frames through ptx_frames_u8_to_ncdhw + the direct stem) against
model.forward (fp32 clips) on the config-2 geometry -- event-timed, table-driven plans.
   usage (GPU box): python scripts/gpu_frames_vs_clips.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ptx = importlib.import_module("pretorched_x_amd")
torch.manual_seed(0)
model = ptx.resnet3d50(num_classes=339, pretrained=None).cuda().eval()
opts = ptx.pretrained_settings["resnet3d50"]["moments"]
frames = torch.randint(0, 256, (8, 16, 224, 224, 3), dtype=torch.uint8, device="cuda")
clips = torch.randn(8, 3, 16, 224, 224, device="cuda")


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


a = timeit(lambda: model(clips))
b = timeit(lambda: model.forward_frames(frames, opts))
print("forward(fp32 clips)   %.4f ms  %.1f clips/s" % (a, 8e3 / a))
print("forward_frames(uint8) %.4f ms  %.1f clips/s   (%+.2f %%)" % (b, 8e3 / b, 100.0 * (b / a - 1.0)))
