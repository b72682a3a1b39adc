"""Model zoo: parameter trees with the reference's weight ABI, executed by the HIP engine.

Each network is described by an `Arch` row and materialised as an `nn.Module` tree whose
`state_dict()` keys, shapes and order are exactly the reference's (SURVEY.md 8b "Weight ABI"):
`conv1.weight`, `bn1.*`, `layer{1-4}.{i}.conv{1,2,3}.weight`, `...bn{1,2,3}.*`,
`...downsample.{0,1}.*`, `last_linear.*` (`fc.*` for R2Plus1D), `...nonlocalblock.{g,theta,phi}.*`,
`...nonlocalblock.W.{0,1}.*`, `....spatial_conv/bn/temporal_conv.*` -- so reference checkpoints
load unchanged.  torch.nn layers are used purely as parameter containers; no torch op computes
anything on the forward path: `features` / `logits` / `forward` run hand-written gfx950 kernels
through `engine.Engine` and raise if that is impossible (CPU tensors, training mode).

Reference behaviour mirrored here (file:line in /root/reference/pretorched/models):
  resnet3D.py:146-218, :242-318 (ResNet3D family + factories), torchvision_models.py:443-492
  (features/logits/forward split, `last_linear`, 2-D resnet18), nonlocalnet.py:423-561
  (NonLocalResNet3D, `num_classes` ignored by `nonlocalresnet3d50`, F8), r2plus1d.py:29-152.
Unlike the reference, methods live on the class of the model itself -- nothing is monkey-patched
onto a shared class.
"""
import math
from dataclasses import dataclass
from typing import Optional, Sequence

import torch
import torch.nn as nn

from . import eager
from .engine import Engine, EngineOwner  # noqa: F401


@dataclass(frozen=True)
class Arch:
    block: str                       # 'basic' | 'bottleneck'
    layers: Sequence[int]
    shortcut: str = "B"
    conv: str = "3d"                 # '3d' | '2p1d'
    nonlocal_layers: Optional[Sequence[int]] = None
    head: str = "last_linear"
    dims: int = 3
    cardinality: int = 32            # 'resnext' blocks only (resnext3D.py:126)
    k: int = 1                       # 'wide' blocks only: width multiplier (wideresnet3D.py:113)

    @property
    def expansion(self):
        return {"bottleneck": 4, "resnext": 2, "wide": 2, "preact_bottleneck": 4}.get(self.block, 1)

    @property
    def widths(self):
        """`planes` of the four stages (resnext3D.py:134-137 doubles them, wideresnet3D.py:125-128 scales by k)."""
        if self.block == "resnext":
            return (128, 256, 512, 1024)
        return tuple(w * self.k for w in (64, 128, 256, 512)) if self.block == "wide" else (64, 128, 256, 512)


ARCHS = {
    "resnet3d10": Arch("basic", (1, 1, 1, 1), "B"),
    "resnet3d18": Arch("basic", (2, 2, 2, 2), "A"),
    "resnet3d34": Arch("basic", (3, 4, 6, 3), "A"),
    "resnet3d50": Arch("bottleneck", (3, 4, 6, 3), "B"),
    "resneti3d50": Arch("bottleneck", (3, 4, 6, 3), "B"),
    "resnet3d101": Arch("bottleneck", (3, 4, 23, 3), "B"),
    "resnet3d152": Arch("bottleneck", (3, 8, 36, 3), "B"),
    "resnet3d200": Arch("bottleneck", (3, 24, 36, 3), "B"),
    "nonlocalresnet3d50": Arch("bottleneck", (3, 4, 6, 3), "A", nonlocal_layers=(0, 2, 3, 0)),
    "r2plus1d10": Arch("basic", (1, 1, 1, 1), "B", conv="2p1d", head="fc"),
    "r2plus1d18": Arch("basic", (2, 2, 2, 2), "B", conv="2p1d", head="fc"),
    "r2plus1d34": Arch("basic", (3, 4, 6, 3), "B", conv="2p1d", head="fc"),
    "r2plus1d50": Arch("bottleneck", (3, 4, 6, 3), "B", conv="2p1d", head="fc"),
    "nonlocal_r2plus1d50": Arch("bottleneck", (3, 4, 6, 3), "B", conv="2p1d", nonlocal_layers=(0, 2, 3, 0)),
    # 2-D torchvision-shaped ResNets (torchvision_models.py:484-536), executed as the T == 1 case;
    # resnet50 is the per-frame backbone of TRN (trn.py:207)
    # ResNeXt3D (resnext3D.py:213-252): grouped 3x3x3 convs, cardinality 32, keeps `fc`, forward only upstream
    "resnext3d10": Arch("resnext", (1, 1, 1, 1), "B", head="fc"),
    "resnext3d18": Arch("resnext", (2, 2, 2, 2), "B", head="fc"),
    "resnext3d34": Arch("resnext", (3, 4, 6, 3), "B", head="fc"),
    "resnext3d50": Arch("resnext", (3, 4, 6, 3), "B", head="fc"),
    "resnext3d101": Arch("resnext", (3, 4, 23, 3), "B", head="fc"),
    "resnext3d152": Arch("resnext", (3, 8, 36, 3), "B", head="fc"),
    "resnext3d200": Arch("resnext", (3, 24, 36, 3), "B", head="fc"),
    # WideResNet-50 3-D (wideresnet3D.py:202-210; module-level upstream, keeps `fc`)
    "wideresnet3d50": Arch("wide", (3, 4, 6, 3), "B", head="fc", k=2),
    # pre-activation ResNet3D (pre_act_resnet3D.py:103-142; module-level upstream, keeps `fc`)
    "preact_resnet3d10": Arch("preact_basic", (1, 1, 1, 1), "B", head="fc"),
    "preact_resnet3d18": Arch("preact_basic", (2, 2, 2, 2), "B", head="fc"),
    "preact_resnet3d34": Arch("preact_basic", (3, 4, 6, 3), "B", head="fc"),
    "preact_resnet3d50": Arch("preact_bottleneck", (3, 4, 6, 3), "B", head="fc"),
    "preact_resnet3d101": Arch("preact_bottleneck", (3, 4, 23, 3), "B", head="fc"),
    "preact_resnet3d152": Arch("preact_bottleneck", (3, 8, 36, 3), "B", head="fc"),
    "preact_resnet3d200": Arch("preact_bottleneck", (3, 24, 36, 3), "B", head="fc"),
    # multi-view ResNets (multiview.py:82-140; module-level upstream, `import resnet3D` absolute, keep `fc`):
    # every conv -- stem, 3x3x3, 1x1x1, shortcut B -- is a MultiViewConv
    "mvresnet10": Arch("basic", (1, 1, 1, 1), "B", conv="mv", head="fc"),
    "mvresnet18": Arch("basic", (2, 2, 2, 2), "B", conv="mv", head="fc"),
    "mvresnet34": Arch("basic", (3, 4, 6, 3), "B", conv="mv", head="fc"),
    "mvresnet50": Arch("bottleneck", (3, 4, 6, 3), "B", conv="mv", head="fc"),
    "mvresnet101": Arch("bottleneck", (3, 4, 23, 3), "B", conv="mv", head="fc"),
    "mvresnet152": Arch("bottleneck", (3, 8, 36, 3), "B", conv="mv", head="fc"),
    "mvresnet200": Arch("bottleneck", (3, 24, 36, 3), "B", conv="mv", head="fc"),
    "resnet18": Arch("basic", (2, 2, 2, 2), "B", dims=2),
    "resnet34": Arch("basic", (3, 4, 6, 3), "B", dims=2),
    "resnet50": Arch("bottleneck", (3, 4, 6, 3), "B", dims=2),
    "resnet101": Arch("bottleneck", (3, 4, 23, 3), "B", dims=2),
    "resnet152": Arch("bottleneck", (3, 8, 36, 3), "B", dims=2),
}


class Bag(nn.Module):
    """A named group of parameter-holding children; never called."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("Bag is a parameter container; the HIP engine executes the network")


def _triple(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


def factored_mid_channels(cin, cout, k):
    """Intermediate width of a (2+1)D pair (reference r2plus1d.py:68-69)."""
    kt, kh, kw = _triple(k)
    return int(math.floor((kt * kh * kw * cin * cout) / (kh * kw * cin + kt * cout)))


class MultiViewConv(nn.Module):
    """reference multiview.py:13-59: ONE 2-D filter bank [Co, Ci, k, k] applied as three 3-D convolutions -- viewed as
    (1,k,k), (k,1,k) and (k,k,1) kernels, each with the matching two-axis padding -- whose outputs are combined by a
    learned Linear(3, 1).  Parameter names / order as upstream (an nn.Conv2d subclass there): `weight`, [`bias`],
    `linear.weight`, `linear.bias`.

    On the HIP engine the three views are ONE dense conv: W3[:, :, t, h, w] = a0 W[h, w] [t == pT] + a1 W[t, w] [h == pH]
    + a2 W[t, h] [w == pW] (a = linear.weight), bias = linear.bias + conv.bias * sum(a) (`effective_weight_bias`, folded at
    pack time like a BatchNorm) -- valid because each view's unpadded axis lines up with tap index == padding, and the
    reference's torch.stack requires 2 * padding == k - 1 anyway (equal output extents)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=False):
        super().__init__()
        if not isinstance(kernel_size, int):
            raise ValueError("MultiViewConv takes one kernel extent, as upstream")
        k = kernel_size
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, 1
        self.kernel_size, self.stride, self.padding = (k, k, k), _triple(stride), _triple(padding)
        if any(2 * p != k - 1 for p in self.padding):
            raise ValueError("MultiViewConv needs 'same' padding (2 * padding == kernel_size - 1): the three views "
                             "must produce equal extents (multiview.py:52-57)")
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, k, k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.linear = nn.Linear(3, 1)

    def views(self):
        k = self.kernel_size[0]
        pt, ph, pw = self.padding
        return [((1, k, k), (0, ph, pw)), ((k, 1, k), (pt, 0, pw)), ((k, k, 1), (pt, ph, 0))]

    def forward(self, x):        # the torch.nn path (train / CPU, eager.py): the reference's own op sequence
        import torch.nn.functional as F
        co, ci = self.out_channels, self.in_channels
        y = torch.stack([F.conv3d(x, self.weight.view(co, ci, *ks), self.bias, self.stride, pad) for ks, pad in self.views()], -1)
        return self.linear(y)[..., 0]

    def effective_weight_bias(self):
        w, a = self.weight.detach(), self.linear.weight.detach()[0]
        k = self.kernel_size[0]
        pt, ph, pw = self.padding
        w3 = torch.zeros(self.out_channels, self.in_channels, k, k, k, device=w.device, dtype=w.dtype)
        w3[:, :, pt, :, :] += a[0] * w
        w3[:, :, :, ph, :] += a[1] * w
        w3[:, :, :, :, pw] += a[2] * w
        b = self.linear.bias.detach().expand(self.out_channels).clone()
        if self.bias is not None:
            b = b + self.bias.detach() * a.sum()
        return w3, b


def _conv(arch, cin, cout, k, stride=1, padding=0, bias=False):
    if arch.dims == 2:
        return nn.Conv2d(cin, cout, k, stride, padding, bias=bias)
    if arch.conv == "mv":
        if not isinstance(k, int):
            k = k[0]
        return MultiViewConv(cin, cout, k, stride, padding, bias=bias)
    if arch.conv == "2p1d":
        (kt, kh, kw), (st, sh, sw), (pt, ph, pw) = _triple(k), _triple(stride), _triple(padding)
        mid = factored_mid_channels(cin, cout, k)
        pair = Bag()
        pair.spatial_conv = nn.Conv3d(cin, mid, (1, kh, kw), (1, sh, sw), (0, ph, pw), bias=bias)
        pair.bn = nn.BatchNorm3d(mid)
        pair.temporal_conv = nn.Conv3d(mid, cout, (kt, 1, 1), (st, 1, 1), (pt, 0, 0), bias=bias)
        return pair
    return nn.Conv3d(cin, cout, k, stride, padding, bias=bias)


def _bn(arch, c):
    return nn.BatchNorm2d(c) if arch.dims == 2 else nn.BatchNorm3d(c)


def _nonlocal(channels):
    inter = max(channels // 2, 1)
    nl = Bag()
    nl.g = nn.Conv3d(channels, inter, 1)
    nl.W = nn.ModuleList([nn.Conv3d(inter, channels, 1), nn.BatchNorm3d(channels)])
    nl.theta = nn.Conv3d(channels, inter, 1)
    nl.phi = nn.Conv3d(channels, inter, 1)
    nl.mode = "embedded_gaussian"
    return nl


class _NonLocalBlockND(EngineOwner, nn.Module):
    """Standalone non-local block with the reference's constructor and parameter names
    (nonlocalnet.py:51-131; NonLocalBlock1D / 2D / 3D :246-270): z = W(y) + x over [B,C,L] / [B,C,H,W] / [B,C,T,H,W].
    HIP path: modes embedded_gaussian, dot_product, gaussian and concatenation, with or without `sub_sample` /
    `bn_layer`.  The block is pointwise convs + attention over positions, so the 1-D and 2-D variants run as the
    T = 1 (and H = 1) case of the same plan; only the `sub_sample` pooling window depends on the dimension."""
    plan_kind = "nlblock"
    dimension = 3

    def __init__(self, in_channels, inter_channels=None, mode="embedded_gaussian", sub_sample=False, bn_layer=True):
        super().__init__()
        assert mode in ["embedded_gaussian", "gaussian", "dot_product", "concatenation"]
        assert self.dimension in (1, 2, 3)
        Conv, Pool, Norm = {3: (nn.Conv3d, nn.MaxPool3d, nn.BatchNorm3d), 2: (nn.Conv2d, nn.MaxPool2d, nn.BatchNorm2d),
                            1: (nn.Conv1d, nn.MaxPool1d, nn.BatchNorm1d)}[self.dimension]
        self.mode, self.sub_sample, self.bn_layer = mode, sub_sample, bn_layer
        self.in_channels = in_channels
        self.inter_channels = inter_channels if inter_channels is not None else max(in_channels // 2, 1)
        ci = self.inter_channels
        self.arch = Arch("nlblock", (), "B")
        self.g = Conv(in_channels, ci, 1)
        if bn_layer:
            self.W = nn.Sequential(Conv(ci, in_channels, 1), Norm(in_channels))
            nn.init.constant_(self.W[1].weight, 0)
            nn.init.constant_(self.W[1].bias, 0)
        else:
            self.W = Conv(ci, in_channels, 1)
            nn.init.constant_(self.W.weight, 0)
            nn.init.constant_(self.W.bias, 0)
        self.theta = self.phi = self.concat_project = None
        if mode in ("embedded_gaussian", "dot_product", "concatenation"):
            self.theta = Conv(in_channels, ci, 1)
            self.phi = Conv(in_channels, ci, 1)
            if mode == "concatenation":
                self.concat_project = nn.Sequential(nn.Conv2d(ci * 2, 1, 1, 1, 0, bias=False), nn.ReLU())
        if sub_sample:
            self.g = nn.Sequential(self.g, Pool(kernel_size=2))
            self.phi = Pool(kernel_size=2) if self.phi is None else nn.Sequential(self.phi, Pool(kernel_size=2))
        self.eval()
        self._init_engine()

    def forward(self, x):
        if eager.wanted(self, x):          # train() / autograd / CPU tensors: the torch.nn path (eager.py)
            eager._count()
            return eager.nonlocal_block(self, x)
        if isinstance(x, torch.Tensor) and x.dim() == self.dimension + 2 and self.dimension < 3:
            b, c = x.shape[:2]
            lead = (1,) * (3 - self.dimension)
            return self._engine.features(self, x.reshape(b, c, *lead, *x.shape[2:])).reshape(x.shape)
        return self._engine.features(self, x)


class NonLocalBlock3D(_NonLocalBlockND):
    dimension = 3


class NonLocalBlock2D(_NonLocalBlockND):
    """nonlocalnet.py:255-261."""
    dimension = 2


class NonLocalBlock1D(_NonLocalBlockND):
    """nonlocalnet.py:246-252."""
    dimension = 1


class MNISTNonLocalNet(EngineOwner, nn.Module):
    """reference nonlocalnet.py:273-309: three conv3x3 (bias) -> BN -> ReLU -> MaxPool2d(2) stages over a 1x28x28 image
    with a NonLocalBlock2D before the second and third conv, then Linear(128*3*3, 256) -> ReLU -> Dropout -> Linear(256, 10).
    Same module tree / state_dict keys (`convs.{0..14}.*`, `fc.{0,3}.*`)."""
    plan_kind = "mnist_nl"

    def __init__(self):
        super().__init__()
        self.arch = Arch("mnist_nl", (), "B", dims=2)
        self.convs = nn.Sequential(
            nn.Conv2d(1, 32, 3, 1, 1), nn.BatchNorm2d(32), nn.ReLU(), nn.MaxPool2d(2),
            NonLocalBlock2D(32), nn.Conv2d(32, 64, 3, 1, 1), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(2),
            NonLocalBlock2D(64), nn.Conv2d(64, 128, 3, 1, 1), nn.BatchNorm2d(128), nn.ReLU(), nn.MaxPool2d(2))
        self.fc = nn.Sequential(nn.Linear(128 * 3 * 3, 256), nn.ReLU(), nn.Dropout(0.5), nn.Linear(256, 10))
        self.eval()
        self._init_engine()

    @property
    def head_module(self):
        return self.fc

    def forward(self, x):
        if eager.wanted(self, x):          # train() / autograd / CPU: the module tree itself (nonlocalnet.py:304-308)
            eager._count()
            return self.fc(self.convs(x).view(x.size(0), -1))
        return self._engine.forward(self, x)


def _block(arch, cin, planes, stride, with_down, with_nl):
    blk = Bag()
    if arch.block == "resnext":          # resnext3D.py:76-99
        mid = arch.cardinality * int(planes / 32)
        blk.conv1 = nn.Conv3d(cin, mid, 1, bias=False)
        blk.bn1 = nn.BatchNorm3d(mid)
        blk.conv2 = nn.Conv3d(mid, mid, 3, stride, 1, groups=arch.cardinality, bias=False)
        blk.bn2 = nn.BatchNorm3d(mid)
        blk.conv3 = nn.Conv3d(mid, planes * 2, 1, bias=False)
        blk.bn3 = nn.BatchNorm3d(planes * 2)
    elif arch.block == "preact_bottleneck":      # pre_act_resnet3D.py:60-74 (BN precedes each conv)
        blk.bn1 = nn.BatchNorm3d(cin)
        blk.conv1 = nn.Conv3d(cin, planes, 1, bias=False)
        blk.bn2 = nn.BatchNorm3d(planes)
        blk.conv2 = nn.Conv3d(planes, planes, 3, stride, 1, bias=False)
        blk.bn3 = nn.BatchNorm3d(planes)
        blk.conv3 = nn.Conv3d(planes, planes * 4, 1, bias=False)
    elif arch.block == "preact_basic":           # pre_act_resnet3D.py:27-39
        blk.bn1 = nn.BatchNorm3d(cin)
        blk.conv1 = nn.Conv3d(cin, planes, 3, stride, 1, bias=False)
        blk.bn2 = nn.BatchNorm3d(planes)
        blk.conv2 = nn.Conv3d(planes, planes, 3, 1, 1, bias=False)
    elif arch.block == "wide":           # wideresnet3D.py:71-84
        blk.conv1 = nn.Conv3d(cin, planes, 1, bias=False)
        blk.bn1 = nn.BatchNorm3d(planes)
        blk.conv2 = nn.Conv3d(planes, planes, 3, stride, 1, bias=False)
        blk.bn2 = nn.BatchNorm3d(planes)
        blk.conv3 = nn.Conv3d(planes, planes * 2, 1, bias=False)
        blk.bn3 = nn.BatchNorm3d(planes * 2)
    elif arch.block == "bottleneck":
        blk.conv1 = _conv(arch, cin, planes, 1)
        blk.bn1 = _bn(arch, planes)
        blk.conv2 = _conv(arch, planes, planes, 3, stride, 1)
        blk.bn2 = _bn(arch, planes)
        blk.conv3 = _conv(arch, planes, planes * 4, 1)
        blk.bn3 = _bn(arch, planes * 4)
    else:
        blk.conv1 = _conv(arch, cin, planes, 3, stride, 1)
        blk.bn1 = _bn(arch, planes)
        blk.conv2 = _conv(arch, planes, planes, 3, 1, 1)
        blk.bn2 = _bn(arch, planes)
    cout = planes * arch.expansion
    blk.downsample = None
    if with_down and arch.shortcut == "B":
        blk.downsample = nn.ModuleList([_conv(arch, cin, cout, 1, stride), _bn(arch, cout)])
    blk.stride = stride
    blk.has_shortcut = with_down
    if with_nl:
        blk.nonlocalblock = _nonlocal(cout)
    blk.has_nl = with_nl
    return blk


def nl_placement(blocks, nonlocal_blocks):
    freq = blocks // nonlocal_blocks if nonlocal_blocks != 0 else -1
    return [(i % freq == 0 and freq > 0) for i in range(blocks)]


class VideoResNet(EngineOwner, nn.Module):
    """ResNet3D / R2Plus1D / NonLocalResNet3D / 2-D ResNet, one class, table driven."""

    def __init__(self, arch_name, num_classes):
        super().__init__()
        arch = ARCHS[arch_name]
        self.arch_name = arch_name
        self.arch = arch
        if arch.dims == 2:
            self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        else:
            self.conv1 = _conv(arch, 3, 64, 7, (1, 2, 2), (3, 3, 3))
        self.bn1 = _bn(arch, 64)
        # parameterless members of the reference trees, kept for structural familiarity only
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = (nn.MaxPool2d(3, 2, 1) if arch.dims == 2 else nn.MaxPool3d(3, 2, 1))
        cin = 64
        for li, (planes, nblocks) in enumerate(zip(arch.widths, arch.layers)):
            stride = 1 if li == 0 else 2
            nl = nl_placement(nblocks, arch.nonlocal_layers[li]) if arch.nonlocal_layers else [False] * nblocks
            blocks = []
            for bi in range(nblocks):
                first = bi == 0
                down = first and (stride != 1 or cin != planes * arch.expansion)
                blocks.append(_block(arch, cin, planes, stride if first else 1, down, nl[bi]))
                if first:
                    cin = planes * arch.expansion
            setattr(self, "layer%d" % (li + 1), nn.ModuleList(blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1) if arch.dims == 2 else nn.AdaptiveAvgPool3d(1)
        if arch.head == "fc":
            self.fc = nn.Linear(arch.widths[3] * arch.expansion, num_classes)
        else:
            self.fc = None                       # the reference sets fc=None after the rename
            self.last_linear = nn.Linear(arch.widths[3] * arch.expansion, num_classes)
        self._init_like_reference()
        self.eval()
        self._init_engine()

    # -- initialisation with the reference's distributions (resnet3D.py:195-201, r2plus1d.py:103) --
    def _init_like_reference(self):
        for m in self.modules():
            if isinstance(m, (nn.Conv3d, nn.Conv2d, MultiViewConv)):          # multiview.py:86-92 for the latter
                nn.init.kaiming_normal_(m.weight, mode="fan_out")
            elif isinstance(m, (nn.BatchNorm3d, nn.BatchNorm2d)):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    @property
    def head_module(self):
        return self.fc if self.arch.head == "fc" else self.last_linear

    # -- the reference Model API (README "Model API"; torchvision_models.py:448-469) --
    # eval mode + ROCm tensors: the HIP engine, always (it raises rather than degrade).  train() mode, inputs
    # that require grad and CPU models are outside its contract and run the torch.nn path (eager.py)
    def features(self, input):
        if eager.wanted(self, input):
            return eager.resnet_features(self, input)
        return self._engine.features(self, input)

    def logits(self, features):
        if eager.wanted(self, features):
            return eager.resnet_logits(self, features)
        return self._engine.logits(self, features)

    def forward(self, input):
        if eager.wanted(self, input):
            return eager.resnet_forward(self, input)
        return self._engine.forward(self, input)

    def forward_frames(self, frames, opts=None):
        """Decoded uint8 frames [B,T,H,W,3] ([B,H,W,3] for the 2-D nets) -> logits: the tensor half of the
        reference's TransformImage (transforms/utils.py:72-75) is fused into the stem's fold kernel.
        opts: mean/std/input_space/input_range holder; default: this model's pretrained settings."""
        return self._engine.forward_frames(self, frames, opts)


# ---------------------------------------------------------------------------------------------
# TRN relation heads (reference trn.py:20-113): standalone modules, HIP-executed MLPs
# ---------------------------------------------------------------------------------------------
class Relation(nn.Module):
    """input[..., num_inputs, in_features] -> output[B, -1, out_features]  (trn.py:20-56)."""

    def __init__(self, num_inputs, in_features, out_features, bottleneck_dim=512):
        super().__init__()
        self.num_inputs, self.in_features = num_inputs, in_features
        self.out_features, self.bottleneck_dim = out_features, bottleneck_dim
        self.relate = nn.Sequential(nn.ReLU(), nn.Linear(num_inputs * in_features, bottleneck_dim),
                                    nn.ReLU(), nn.Linear(bottleneck_dim, out_features))
        self.eval()

    def forward(self, input):
        from .engine import relation_mlp
        if eager.wanted(self, input):
            return eager.relation(self, input)
        flat = input.contiguous().view(-1, self.num_inputs * self.in_features)
        out = relation_mlp(flat, self.relate[1], self.relate[3])
        return out.view(input.size(0), -1, self.out_features)


class MultiScaleRelation(nn.Module):
    """Sum of k-frame relations for k = n..2 over randomly drawn frame subsets (trn.py:59-113).
    Subset sampling stays on the host and consumes numpy's global RNG exactly as the reference
    does (one `np.random.choice` per scale, in scale order); only the MLPs run on the GPU."""

    def __init__(self, num_input, in_features, out_features, bottleneck_dim=512, num_relations=3):
        super().__init__()
        import itertools
        self.num_input, self.in_features, self.out_features = num_input, in_features, out_features
        self.num_relations, self.bottleneck_dim = num_relations, bottleneck_dim
        self.scales = list(range(num_input, 1, -1))
        self.relations_scales = [list(itertools.combinations(range(num_input), s)) for s in self.scales]
        self.subsample_scales = [min(num_relations, len(r)) for r in self.relations_scales]
        self.relations = nn.ModuleList([Relation(s, in_features, out_features, bottleneck_dim)
                                        for s in self.scales])
        self.eval()

    def forward(self, input):
        import numpy as np
        from ._lib import PTX_REL_MAX_FRAMES, PTX_REL_MAX_SETS
        from .engine import relation_mlp, relation_scale
        x = input.contiguous().view(-1, self.num_input, self.in_features)        # [B, T, F]
        if eager.wanted(self, input):          # trn.py:95-113 on the torch.nn path: same RNG consumption
            outs = []
            for si in range(len(self.scales)):
                picks = np.random.choice(len(self.relations_scales[si]), self.subsample_scales[si], replace=False)
                for idx in picks:
                    outs.append(self.relations[si](input[..., self.relations_scales[si][idx], :]))
            return torch.stack(outs).sum(0).view(input.size(0), -1, self.out_features)
        grouped = (self.in_features % 4 == 0 and self.num_input <= PTX_REL_MAX_FRAMES
                   and max(self.subsample_scales) <= PTX_REL_MAX_SETS)
        total = None
        for si in range(len(self.scales)):
            rel = self.relations[si]
            picks = np.random.choice(len(self.relations_scales[si]), self.subsample_scales[si], replace=False)
            subsets = [self.relations_scales[si][idx] for idx in picks]
            if grouped:
                # every subset of this scale in two launches: frames are gathered inside the kernel, W1 is
                # read once, W2 acts on the summed hidden vectors (trn.py:110 by linearity)
                total = relation_scale(x, subsets, rel.relate[1], rel.relate[3], out=total, accumulate=total is not None)
                continue
            for sub_idx in subsets:
                flat = x[:, sub_idx, :].contiguous().view(-1, rel.num_inputs * rel.in_features)
                total = relation_mlp(flat, rel.relate[1], rel.relate[3], out=total, accumulate=total is not None)
        return total.view(input.size(0), -1, self.out_features)


class HierarchicalRelation(nn.Module):
    """trn.py:115-160.  The reference's forward only runs when `depth == 0` (relation_size >=
    num_inputs): for depth >= 1 its `linear(input).sum(-2)` reduces the singleton axis left by
    `Relation.view(B, -1, out)`, the per-level outputs keep different window counts and
    `torch.stack(outs)` raises.  TRN always lands on depth 0, because it passes
    `frame_bottleneck_dim` (1024) in the `relation_size` slot (trn.py:230-233).  Mirrored: the
    parameter tree (relations / linears / final_linear / final_relation) is built for any depth so
    checkpoints load; forward executes depth 0 on the GPU and raises the reference's error class
    otherwise."""

    def __init__(self, num_inputs, in_features, out_features, relation_size=4, relation_dist=1,
                 bottleneck_dim=1024):
        super().__init__()
        import math
        self.num_inputs, self.in_features, self.out_features = num_inputs, in_features, out_features
        self.relation_size, self.relation_dist, self.bottleneck_dim = relation_size, relation_dist, bottleneck_dim
        depth = int(math.ceil((num_inputs - relation_size) / (relation_size - 1)))
        self.depth = max(depth, 0)
        num_inputs_final = num_inputs + depth * (1 - relation_size)
        self.relations = nn.ModuleList([Relation(relation_size, in_features, in_features)
                                        for _ in range(self.depth)])
        self.linears = nn.ModuleList([nn.Linear(in_features, out_features) for _ in range(self.depth)])
        self.final_linear = nn.Linear(in_features, out_features)       # unused by forward upstream too
        self.final_relation = Relation(num_inputs_final, in_features, out_features)
        self.eval()

    def forward(self, input):
        if self.depth != 0:
            raise RuntimeError("HierarchicalRelation with depth %d: the reference forward (trn.py:150-160) "
                               "raises in torch.stack for depth >= 1; only depth 0 is defined" % self.depth)
        input = input.view(-1, self.num_inputs, self.in_features)
        # stack([final_relation(input)]).mean(0) over one element is the identity
        return self.final_relation(input)


class MultiScaleHierarchicalRelation(nn.Module):
    """trn.py:163-191: parameter tree only.  Every scale < num_inputs builds a depth >= 1
    HierarchicalRelation, whose reference forward raises (see above), so no output is defined."""

    def __init__(self, num_inputs, in_features, out_features, relation_dist=1, bottleneck_dim=512):
        super().__init__()
        self.num_inputs, self.in_features, self.out_features = num_inputs, in_features, out_features
        self.scales = range(num_inputs, 1, -1)
        self.num_scales = len(self.scales)
        self.h_relations = nn.ModuleList([
            HierarchicalRelation(num_inputs, in_features, out_features, relation_size=s,
                                 relation_dist=relation_dist, bottleneck_dim=bottleneck_dim)
            for s in self.scales])
        self.final_relation = Relation(self.num_scales, out_features, out_features, bottleneck_dim=bottleneck_dim)
        self.eval()

    def forward(self, input):
        raise RuntimeError("MultiScaleHierarchicalRelation: the reference forward (trn.py:186-191) raises "
                           "in torch.stack for every num_inputs > 2; no output is defined")


class TRN(nn.Module):
    """Temporal Relation Network (trn.py:194-263): a 2-D backbone run on every frame, a temporal
    relation head over the per-frame features and a Linear classifier.  All arithmetic runs in
    libptx_amd: frames go through the HIP engine as a [B*T,3,H,W] batch (the backbone's
    `last_linear` is a Dropout, i.e. the identity here), the relation MLPs and the classifier
    through ptx_linear_fwd.

    `pretrained=None` builds the backbone without weights (the reference cannot: it reads
    `base_model.mean/std`, which only exist after a download, trn.py:211-214); the settings are
    then taken from the arch's 'moments' (else 'imagenet') entry."""

    consensus_mods = {"TRN": Relation, "HTRN": HierarchicalRelation, "MSTRN": MultiScaleRelation,
                      "MSHTRN": MultiScaleHierarchicalRelation}

    def __init__(self, num_classes, num_segments=8, arch="resnet50", frame_bottleneck_dim=1024,
                 video_feature_dim=1024, consensus="HTRN", pretrained="moments", dropout=0.5, partial_bn=True):
        super().__init__()
        from . import __dict__ as factories, pretrained_settings
        self.arch, self.reshape, self.dropout = arch, True, dropout
        self.consensus, self.num_classes, self.num_segments = consensus, num_classes, num_segments
        self.video_feature_dim, self.frame_bottleneck_dim = video_feature_dim, frame_bottleneck_dim
        num_pc = 1000 if pretrained == "imagenet" else 339
        self.base_model = factories[arch](num_pc, pretrained)
        if pretrained is None:
            table = pretrained_settings[arch]
            settings = table.get("moments", table.get("imagenet"))
            for k in ("input_space", "input_size", "input_range", "mean", "std"):
                setattr(self.base_model, k, settings[k])
        self.frame_feature_dim = self.base_model.last_linear.in_features
        self.base_model.last_linear = nn.Dropout(self.dropout)
        self.std, self.mean = self.base_model.std, self.base_model.mean
        self.input_size = self.base_model.input_size[1:]
        self.input_space = self.base_model.input_space
        if consensus not in self.consensus_mods:
            raise ValueError("Unrecognized temporal consensus.")
        self.temporal_relation = self.consensus_mods[consensus](
            self.num_segments, self.frame_feature_dim, self.video_feature_dim, self.frame_bottleneck_dim)
        self.last_linear = nn.Linear(self.video_feature_dim, self.num_classes)
        self.eval()

    def features(self, input):
        """[B, T, 3, H, W] (or [B, T*3, H, W]) -> [B, video_feature_dim]  (trn.py:246-255);
        like the reference, `.squeeze()` also drops the batch axis when B == 1."""
        batch_size = input.size(0)
        frames = input.reshape((-1, 3) + tuple(input.shape[-2:]))
        base_rep = self.base_model(frames)                                  # [B*T, F]
        base_rep = base_rep.view(batch_size, -1, self.num_segments, base_rep.size(-1))
        return self.temporal_relation(base_rep).squeeze()

    def logits(self, features):
        from .engine import linear
        return linear(features, self.last_linear)

    def forward(self, input):
        return self.logits(self.features(input))

    @property
    def crop_size(self):
        return self.input_size

    @property
    def scale_size(self):
        return self.input_size[0] * 256 // 224
