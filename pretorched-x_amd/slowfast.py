"""SlowFast networks behind the HIP engine (reference pretorched/models/slowfast.py, registered as
`pretorched.slowfast` at pretorched/__init__.py:83).

Same constructors and `state_dict` ABI as the reference classes -- `SlowFast` (`slow.*`, `fast.*`,
`fast.lateral_*`, bias-free `last_linear`), `SlowOnly`, `FastOnly` and the `resnet18/50/101/152/200`
factories with their `mode='SF'|'S'|'F'` switch -- but the modules only hold parameters: the forward
pass is compiled by `engine.Plan._build_slowfast` into libptx_amd launches.

What the engine does differently from the reference graph (slowfast.py:140-156, 280-299, 385-398):
  * `input[:, :, ::stride]` is a frame stride of the stem's fold kernel, not a strided copy;
  * `torch.cat([x, lateral], dim=1)` never happens: the stage's last conv and the lateral conv write
    their channel slices of one pre-allocated tensor (conv/pool outputs take a row stride);
  * BN is folded into the filters, ReLU / residual adds run in the conv epilogue, the stage-entry
    shortcut conv is K-concatenated into conv3's GEMM.
"""
import torch.nn as nn

from .engine import EngineOwner
from .zoo import Arch, Bag

# the reference passes block *classes*; here the two kinds are named by these constants
BasicBlock, Bottleneck = "basic", "bottleneck"

__all__ = ["SlowFast", "SlowOnly", "FastOnly", "resnet18", "resnet50", "resnet101", "resnet152", "resnet200"]


def _block(kind, inplanes, planes, stride=1, downsample=None, head_conv=1):
    """Parameter tree of slowfast.BasicBlock (:8-53) / slowfast.Bottleneck (:56-99)."""
    blk = Bag()
    if kind == "basic":
        if head_conv == 1:
            blk.conv1 = nn.Conv3d(inplanes, planes, (1, 3, 3), (1, stride, stride), (0, 1, 1), bias=False)
        elif head_conv == 3:
            blk.conv1 = nn.Conv3d(inplanes, planes, (3, 1, 1), padding=(1, 0, 0), bias=False)
        else:
            raise ValueError("Unsupported head_conv")
        blk.bn1 = nn.BatchNorm3d(planes)
        blk.conv2 = nn.Conv3d(planes, planes, (1, 3, 3), (1, stride, stride), (0, 1, 1))   # bias=True upstream
        blk.bn2 = nn.BatchNorm3d(planes)
        blk.out_channels = planes
    else:
        if head_conv == 1:
            blk.conv1 = nn.Conv3d(inplanes, planes, 1, bias=False)
        elif head_conv == 3:
            blk.conv1 = nn.Conv3d(inplanes, planes, (3, 1, 1), bias=False, padding=(1, 0, 0))
        else:
            raise ValueError("Unsupported head_conv!")
        blk.bn1 = nn.BatchNorm3d(planes)
        blk.conv2 = nn.Conv3d(planes, planes, (1, 3, 3), (1, stride, stride), (0, 1, 1), bias=False)
        blk.bn2 = nn.BatchNorm3d(planes)
        blk.conv3 = nn.Conv3d(planes, planes * 4, 1, bias=False)
        blk.bn3 = nn.BatchNorm3d(planes * 4)
        blk.out_channels = planes * 4
    blk.downsample = downsample
    blk.stride = stride
    blk.has_shortcut = downsample is not None
    blk.has_nl = False
    return blk


class _Pathway(nn.Module):
    """Shared construction of the Slow (:102-196) and Fast (:244-327) pathways."""

    def _init_pathway(self, block, layers):
        if block not in ("basic", "bottleneck"):
            raise ValueError("block must be 'basic' or 'bottleneck'")
        self.block = block
        self.expansion = 4 if block == "bottleneck" else 1
        self.layers = tuple(layers)
        self.arch = Arch(block, self.layers, "B")

    def _make_layer(self, planes, blocks, stride=1, head_conv=1, lateral_growth=False):
        exp = self.expansion
        downsample = None
        if stride != 1 or self.inplanes != planes * exp:
            downsample = nn.ModuleList([nn.Conv3d(self.inplanes, planes * exp, 1, (1, stride, stride), bias=False),
                                        nn.BatchNorm3d(planes * exp)])
        out = [_block(self.block, self.inplanes, planes, stride, downsample, head_conv)]
        self.inplanes = planes * exp
        for _ in range(1, blocks):
            out.append(_block(self.block, self.inplanes, planes, head_conv=head_conv))
        if lateral_growth:      # the next slow stage also sees the fast pathway's lateral features
            self.inplanes = planes * exp + planes * exp // 8 * 2
        return nn.ModuleList(out)

    def _make_slow_layers(self, lateral):
        layers = self.layers
        self.conv1 = nn.Conv3d(3, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        res3_stride = 2 if self.block == "bottleneck" else 1
        self.res2 = self._make_layer(64, layers[0], head_conv=1, lateral_growth=lateral)
        self.res3 = self._make_layer(128, layers[1], stride=res3_stride, head_conv=1, lateral_growth=lateral)
        self.res4 = self._make_layer(256, layers[2], stride=2, head_conv=3, lateral_growth=lateral)
        self.res5 = self._make_layer(512, layers[3], stride=2, head_conv=3, lateral_growth=lateral)

    def _make_fast_layers(self, lateral):
        layers, exp = self.layers, self.expansion
        self.inplanes = 8
        self.conv1 = nn.Conv3d(3, 8, (5, 7, 7), (1, 2, 2), (2, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(8)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d((1, 3, 3), (1, 2, 2), (0, 1, 1))
        res3_stride = 2 if self.block == "bottleneck" else 1
        self.res2 = self._make_layer(8, layers[0], head_conv=3)
        self.res3 = self._make_layer(16, layers[1], stride=res3_stride, head_conv=3)
        self.res4 = self._make_layer(32, layers[2], stride=2, head_conv=3)
        self.res5 = self._make_layer(64, layers[3], stride=2, head_conv=3)
        if lateral:
            def lat(c):
                return nn.Conv3d(c, c * 2, (5, 1, 1), (8, 1, 1), (2, 0, 0), bias=False)
            self.lateral_p1 = lat(8)
            self.lateral_res2 = lat(8 * exp)
            self.lateral_res3 = lat(16 * exp)
            self.lateral_res4 = lat(32 * exp)


class Slow(_Pathway):
    """slowfast.Slow (:102-196): parameters of the slow pathway inside SlowFast."""

    def __init__(self, block="bottleneck", layers=(2, 2, 2, 2)):
        super().__init__()
        self._init_pathway(block, layers)
        self.inplanes = 64 + 64 // 8 * 2
        self._make_slow_layers(lateral=True)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("Slow holds parameters; run it through SlowFast or SlowOnly")


class Fast(_Pathway):
    """slowfast.Fast (:244-327): parameters of the fast pathway (+ lateral convs) inside SlowFast."""

    def __init__(self, block="bottleneck", layers=(2, 2, 2, 2)):
        super().__init__()
        self._init_pathway(block, layers)
        self._make_fast_layers(lateral=True)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("Fast holds parameters; run it through SlowFast or FastOnly")


class _Runnable(EngineOwner):
    plan_kind = "slowfast"

    @property
    def head_module(self):
        return self.last_linear

    def forward(self, input):
        """[B,3,T,H,W] fp32 CUDA clip -> [B,num_classes]  (dropout is the identity in eval mode)."""
        return self._engine.forward(self, input)

    def forward_frames(self, frames, opts):
        """Decoded uint8 frames [B,T,H,W,3] -> logits, normalisation fused into the stems."""
        return self._engine.forward_frames(self, frames, opts)


class SlowFast(_Runnable, nn.Module):
    """slowfast.SlowFast (:366-398)."""
    mode = "sf"

    def __init__(self, block="bottleneck", layers=(2, 2, 2, 2), num_classes=400, dropout=0.5, slow_stride=16,
                 fast_stride=2):
        super().__init__()
        self.slow_stride, self.fast_stride = slow_stride, fast_stride
        self.slow = Slow(block, layers)
        self.fast = Fast(block, layers)
        self.expansion = self.slow.expansion
        self.arch = self.slow.arch
        self.dropout = nn.Dropout(dropout)
        self.last_linear = nn.Linear(self.fast.inplanes + 512 * self.expansion, num_classes, bias=False)
        self.eval()
        self._init_engine()


class SlowOnly(_Runnable, _Pathway):
    """slowfast.SlowOnly (:199-241): the slow pathway alone, frames subsampled by `slow_stride`."""
    mode = "s"

    def __init__(self, block="bottleneck", layers=(2, 2, 2, 2), num_classes=400, dropout=0.5, slow_stride=16):
        nn.Module.__init__(self)
        self._init_pathway(block, layers)
        self.inplanes = 64
        self.slow_stride = slow_stride
        self._make_slow_layers(lateral=False)
        self.dropout = nn.Dropout(dropout)
        self.last_linear = nn.Linear(self.inplanes, num_classes)
        self.eval()
        self._init_engine()


class FastOnly(_Runnable, _Pathway):
    """slowfast.FastOnly (:330-363): the fast pathway alone, frames subsampled by `fast_stride`."""
    mode = "f"

    def __init__(self, block="bottleneck", layers=(2, 2, 2, 2), num_classes=400, dropout=0.5, fast_stride=2):
        nn.Module.__init__(self)
        self._init_pathway(block, layers)
        self.fast_stride = fast_stride
        self._make_fast_layers(lateral=False)
        self.dropout = nn.Dropout(dropout)
        self.last_linear = nn.Linear(self.inplanes, num_classes)
        self.eval()
        self._init_engine()


_MODES = {"sf": SlowFast, "f": FastOnly, "s": SlowOnly}


def _pick(mode):
    try:
        return _MODES[mode.lower()]
    except KeyError:
        # upstream: `models.get(mode.lower(), 'sf')` returns the *string* 'sf' and the call fails
        raise TypeError("unknown SlowFast mode %r (use 'SF', 'S' or 'F')" % (mode,))


def resnet18(mode="SF", **kwargs):
    """slowfast.py:578-583"""
    return _pick(mode)("basic", [2, 2, 2, 2], **kwargs)


def resnet50(mode="SF", **kwargs):
    """slowfast.py:586-591"""
    return _pick(mode)("bottleneck", [3, 4, 6, 3], **kwargs)


def resnet101(**kwargs):
    """slowfast.py:594-598"""
    return SlowFast("bottleneck", [3, 4, 23, 3], **kwargs)


def resnet152(**kwargs):
    """slowfast.py:601-605"""
    return SlowFast("bottleneck", [3, 8, 36, 3], **kwargs)


def resnet200(**kwargs):
    """slowfast.py:608-612"""
    return SlowFast("bottleneck", [3, 24, 36, 3], **kwargs)
