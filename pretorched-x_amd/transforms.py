"""Pre-processing edge on the device (reference pretorched/transforms/utils.py:34-81).

`TransformImage` there is PIL resize/crop (host work, not on this path) followed by a tensor half:
ToTensor (uint8 HWC -> float CHW / 255), ToSpaceBGR, ToRange255, Normalize(mean, std).  For video
that tensor half runs per frame and the frames are stacked into [3,T,H,W]; `FramesToTensor` does
the same arithmetic (same fp32 operations in the same order: bit-identical) for a whole batch of
decoded uint8 frames in one HIP launch.  Models go one step further with `model.forward_frames`,
which fuses it into the stem's fold kernel so the fp32 clip never exists in HBM.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import NormDesc, PtxError, check


def _opt(opts, key):
    return opts[key] if isinstance(opts, dict) else getattr(opts, key)


class FramesToTensor:
    """opts: a model (after `pretrained=...`) or a `pretrained_settings[...]` dict -- anything with
    input_space / input_range / mean / std, as TransformImage takes (utils.py:36-45)."""

    def __init__(self, opts):
        self.input_space, self.input_range = _opt(opts, "input_space"), _opt(opts, "input_range")
        self.mean, self.std = list(_opt(opts, "mean")), list(_opt(opts, "std"))
        self.norm = NormDesc.make(self.mean, self.std, self.input_space, self.input_range)

    def __call__(self, frames):
        """uint8 CUDA frames [N,T,H,W,C] | [T,H,W,C] | [H,W,C]  ->  fp32 [N,C,T,H,W] | [C,T,H,W] | [C,H,W]."""
        if not isinstance(frames, torch.Tensor) or not frames.is_cuda or frames.dtype != torch.uint8:
            raise PtxError("FramesToTensor: frames must be a uint8 CUDA tensor (no CPU fallback)")
        if frames.dim() not in (3, 4, 5):
            raise PtxError("FramesToTensor: expected [N,T,H,W,C], [T,H,W,C] or [H,W,C]")
        lead = frames.dim()
        f5 = frames.contiguous().view((1,) * (5 - lead) + tuple(frames.shape))
        N, T, H, W, Cc = f5.shape
        with torch.cuda.device(frames.device):
            out = torch.empty((N, Cc, T, H, W), device=frames.device, dtype=torch.float32)
            check(_lib.lib().ptx_frames_u8_to_ncdhw(C.c_void_p(f5.data_ptr()), C.c_void_p(out.data_ptr()), N, T, H, W,
                                                    Cc, C.byref(self.norm),
                                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                  "ptx_frames_u8_to_ncdhw")
        if lead == 5:
            return out
        return out[0] if lead == 4 else out[0, :, 0]
