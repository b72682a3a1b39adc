"""TEST INFRASTRUCTURE ONLY -- functional CPU restatement of the reference forward pass.

Everything here is a pure function of (state_dict, input): weights are looked up by the
reference's own `state_dict` key names (the weight ABI), and the ATen ops
are issued in the order the reference's nn.Modules issue them, so on the same machine the results
are bit-identical to the imported reference (asserted by tests/test_oracle_vs_reference.py).

Reference lines followed:
  * stem / stages / head ............ pretorched/models/resnet3D.py:146-218,
                                      pretorched/models/torchvision_models.py:443-469
  * BasicBlock / Bottleneck ......... resnet3D.py:77-143
  * shortcut A ...................... resnet3D.py:65-74, nonlocalnet.py:322-332
  * shortcut B ...................... resnet3D.py:175-185
  * (2+1)D factored conv ............ r2plus1d.py:29-88 (mid-channel formula :68-69)
  * non-local block (4 modes) ....... nonlocalnet.py:51-243
  * NL-ResNet3D placement rule ...... nonlocalnet.py:456-485
  * TRN relation MLP ................ trn.py:20-56, multi-scale :59-113
  * 2-D ResNet (torchvision shape) .. torchvision_models.py:443-492 + oracle/tv_standin.py
  * MultiViewConv / MVResNet ........ multiview.py:13-59, :82-140
"""
import itertools
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5  # nn.BatchNorm{2,3}d default, never overridden by the reference


@dataclass
class ArchCfg:
    """Static description of one ResNet-style network of the reference zoo."""
    block: str                      # 'basic' | 'bottleneck'
    layers: Sequence[int]
    shortcut: str = "B"            # 'A' zero-pad (resnet3D.py:65) | 'B' conv+bn (resnet3D.py:175)
    conv: str = "3d"               # '3d' plain nn.Conv3d | '2p1d' SpatioTemporalConv everywhere
    nonlocal_layers: Optional[Sequence[int]] = None   # NonLocalResNet3D only
    nl_mode: str = "embedded_gaussian"
    head: str = "last_linear"      # 'fc' for R2Plus1D (never passes through modify_resnets)
    dims: int = 3                  # 2 for the torchvision-shaped resnet18 plumbing case
    cardinality: int = 32          # 'resnext' blocks (resnext3D.py:126)
    k: int = 1                     # 'wide' blocks (wideresnet3D.py:113)
    expansion: int = field(init=False)

    def __post_init__(self):
        self.expansion = {"bottleneck": 4, "resnext": 2, "wide": 2, "preact_bottleneck": 4}.get(self.block, 1)

    @property
    def widths(self):
        if self.block == "resnext":
            return (128, 256, 512, 1024)
        return tuple(w * self.k for w in (64, 128, 256, 512)) if self.block == "wide" else (64, 128, 256, 512)


ARCHS = {
    "resnet3d10": ArchCfg("basic", [1, 1, 1, 1], "B"),
    "resnet3d18": ArchCfg("basic", [2, 2, 2, 2], "A"),
    "resnet3d34": ArchCfg("basic", [3, 4, 6, 3], "A"),
    "resnet3d50": ArchCfg("bottleneck", [3, 4, 6, 3], "B"),
    "resneti3d50": ArchCfg("bottleneck", [3, 4, 6, 3], "B"),
    "resnet3d101": ArchCfg("bottleneck", [3, 4, 23, 3], "B"),
    "resnet3d152": ArchCfg("bottleneck", [3, 8, 36, 3], "B"),
    "resnet3d200": ArchCfg("bottleneck", [3, 24, 36, 3], "B"),
    "nonlocalresnet3d50": ArchCfg("bottleneck", [3, 4, 6, 3], "A", nonlocal_layers=[0, 2, 3, 0]),
    "r2plus1d10": ArchCfg("basic", [1, 1, 1, 1], "B", conv="2p1d", head="fc"),
    "r2plus1d18": ArchCfg("basic", [2, 2, 2, 2], "B", conv="2p1d", head="fc"),
    "r2plus1d34": ArchCfg("basic", [3, 4, 6, 3], "B", conv="2p1d", head="fc"),
    "r2plus1d50": ArchCfg("bottleneck", [3, 4, 6, 3], "B", conv="2p1d", head="fc"),
    # config-3 composite: (2+1)D bottlenecks + NL blocks, shortcut B
    "nonlocal_r2plus1d50": ArchCfg("bottleneck", [3, 4, 6, 3], "B", conv="2p1d",
                                   nonlocal_layers=[0, 2, 3, 0]),
    "resnext3d10": ArchCfg("resnext", [1, 1, 1, 1], "B", head="fc"),
    "resnext3d18": ArchCfg("resnext", [2, 2, 2, 2], "B", head="fc"),
    "resnext3d50": ArchCfg("resnext", [3, 4, 6, 3], "B", head="fc"),
    "resnext3d101": ArchCfg("resnext", [3, 4, 23, 3], "B", head="fc"),
    "wideresnet3d50": ArchCfg("wide", [3, 4, 6, 3], "B", head="fc", k=2),
    "preact_resnet3d10": ArchCfg("preact_basic", [1, 1, 1, 1], "B", head="fc"),
    "preact_resnet3d18": ArchCfg("preact_basic", [2, 2, 2, 2], "B", head="fc"),
    "preact_resnet3d50": ArchCfg("preact_bottleneck", [3, 4, 6, 3], "B", head="fc"),
    "mvresnet10": ArchCfg("basic", [1, 1, 1, 1], "B", conv="mv", head="fc"),
    "mvresnet18": ArchCfg("basic", [2, 2, 2, 2], "B", conv="mv", head="fc"),
    "mvresnet34": ArchCfg("basic", [3, 4, 6, 3], "B", conv="mv", head="fc"),
    "mvresnet50": ArchCfg("bottleneck", [3, 4, 6, 3], "B", conv="mv", head="fc"),
    "resnet18": ArchCfg("basic", [2, 2, 2, 2], "B", dims=2),
    "resnet34": ArchCfg("basic", [3, 4, 6, 3], "B", dims=2),
    "resnet50": ArchCfg("bottleneck", [3, 4, 6, 3], "B", dims=2),
    "resnet101": ArchCfg("bottleneck", [3, 4, 23, 3], "B", dims=2),
    "resnet152": ArchCfg("bottleneck", [3, 8, 36, 3], "B", dims=2),
}


def _t3(v):
    return (v, v, v) if isinstance(v, int) else tuple(v)


# --------------------------------------------------------------------------------------------
# primitive pieces
# --------------------------------------------------------------------------------------------
def _bn(sd, x, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.1, BN_EPS)


def _conv(sd, x, p, stride, padding, dims=3):
    w = sd[p + ".weight"]
    b = sd.get(p + ".bias")
    if dims == 2:
        return F.conv2d(x, w, b, stride, padding)
    return F.conv3d(x, w, b, _t3(stride), _t3(padding))


def st_mid_channels(cin, cout, k):
    """r2plus1d.py:68-69."""
    kt, kh, kw = _t3(k)
    return int(math.floor((kt * kh * kw * cin * cout) / (kh * kw * cin + kt * cout)))


def _st_conv(sd, x, p, stride, padding):
    """SpatioTemporalConv.forward, r2plus1d.py:85-88."""
    st, sh, sw = _t3(stride)
    pt, ph, pw = _t3(padding)
    x = _conv(sd, x, p + ".spatial_conv", (1, sh, sw), (0, ph, pw))
    x = F.relu(_bn(sd, x, p + ".bn"))
    return _conv(sd, x, p + ".temporal_conv", (st, 1, 1), (pt, 0, 0))


def _mv_conv(sd, x, p, stride, padding):
    """MultiViewConv.forward, multiview.py:51-59: the 2-D bank viewed as (1,k,k) / (k,1,k) / (k,k,1) kernels, each with
    the padding of its two live axes, stacked on a new last axis and combined by Linear(3, 1)."""
    w = sd[p + ".weight"]
    b = sd.get(p + ".bias")
    co, ci, k = w.shape[0], w.shape[1], w.shape[2]
    pt, ph, pw = _t3(padding)
    views = [((1, k, k), (0, ph, pw)), ((k, 1, k), (pt, 0, pw)), ((k, k, 1), (pt, ph, 0))]
    y = torch.stack([F.conv3d(x, w.view(co, ci, *ks), b, _t3(stride), pad, (1, 1, 1), 1) for ks, pad in views], -1)
    return F.linear(y, sd[p + ".linear.weight"], sd[p + ".linear.bias"])[..., 0]


def _any_conv(cfg, sd, x, p, stride, padding):
    if cfg.conv == "mv":
        return _mv_conv(sd, x, p, stride, padding)
    if cfg.conv == "2p1d":
        return _st_conv(sd, x, p, stride, padding)
    return _conv(sd, x, p, stride, padding, cfg.dims)


def shortcut_a(x, planes, stride):
    """downsample_basic_block: strided subsample (avg_pool k=1) + zero channels."""
    out = F.avg_pool3d(x, kernel_size=1, stride=stride)
    pad = torch.zeros(out.size(0), planes - out.size(1), *out.shape[2:], dtype=out.dtype)
    return torch.cat([out, pad], dim=1)


# --------------------------------------------------------------------------------------------
# non-local block
# --------------------------------------------------------------------------------------------
def nonlocal_block(sd, x, p, mode="embedded_gaussian", sub_sample=False, bn_layer=True):
    """_NonLocalBlockND (nonlocalnet.py:139-243); the dimension (1 / 2 / 3, :246-270) is read off the input."""
    b, c = x.shape[:2]
    gk = p + ".g.0" if sub_sample else p + ".g"
    ci = sd[gk + ".weight"].shape[0]
    conv_nd = {3: F.conv1d, 4: F.conv2d, 5: F.conv3d}[x.dim()]
    pool_nd = {3: F.max_pool1d, 4: F.max_pool2d, 5: F.max_pool3d}[x.dim()]

    def pw(key, inp):
        return conv_nd(inp, sd[key + ".weight"], sd.get(key + ".bias"))

    def pool(t):
        return pool_nd(t, 2) if sub_sample else t

    g_x = pool(pw(gk, x)).reshape(b, ci, -1).permute(0, 2, 1)
    if mode == "gaussian":
        theta_x = x.reshape(b, c, -1).permute(0, 2, 1)
        phi_x = pool(x).reshape(b, c, -1)
        f = F.softmax(torch.matmul(theta_x, phi_x), dim=-1)
    else:
        pk = p + ".phi.0" if sub_sample else p + ".phi"
        theta = pw(p + ".theta", x)
        phi = pool(pw(pk, x))
        if mode == "concatenation":
            th = theta.reshape(b, ci, -1, 1)
            ph = phi.reshape(b, ci, 1, -1)
            h, w = th.size(2), ph.size(3)
            cat = torch.cat([th.repeat(1, 1, 1, w), ph.repeat(1, 1, h, 1)], dim=1)
            f = F.relu(F.conv2d(cat, sd[p + ".concat_project.0.weight"]))
            f = f.reshape(b, h, w)
            f = f / f.size(-1)
        else:
            f = torch.matmul(theta.reshape(b, ci, -1).permute(0, 2, 1), phi.reshape(b, ci, -1))
            f = F.softmax(f, dim=-1) if mode == "embedded_gaussian" else f / f.size(-1)
    y = torch.matmul(f, g_x).permute(0, 2, 1).contiguous().reshape(b, ci, *x.shape[2:])
    if bn_layer:
        w_y = _bn(sd, pw(p + ".W.0", y), p + ".W.1")
    else:
        w_y = pw(p + ".W", y)
    return w_y + x


def mnist_nonlocal_forward(sd, x):
    """MNISTNonLocalNet.forward (nonlocalnet.py:273-309): `convs` = [conv3x3(bias), BN, ReLU, MaxPool2d(2)] x 3 with a
    NonLocalBlock2D at indices 4 and 9; `fc` = Linear, ReLU, Dropout (identity in eval), Linear."""
    with torch.no_grad():
        for i in (0, 5, 10):
            p = "convs.%d" % i
            x = F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], 1, 1)
            q = "convs.%d" % (i + 1)
            x = F.batch_norm(x, sd[q + ".running_mean"], sd[q + ".running_var"], sd[q + ".weight"], sd[q + ".bias"], False, 0.1, BN_EPS)
            x = F.max_pool2d(F.relu(x), 2)
            if i < 10:
                x = nonlocal_block(sd, x, "convs.%d" % (i + 4))
        x = x.view(x.size(0), -1)
        x = F.relu(F.linear(x, sd["fc.0.weight"], sd["fc.0.bias"]))
        return F.linear(x, sd["fc.3.weight"], sd["fc.3.bias"])


# --------------------------------------------------------------------------------------------
# residual stages
# --------------------------------------------------------------------------------------------
def _preact_block(cfg, sd, x, p, planes, stride, has_down):
    """PreActivationBottleneck.forward (pre_act_resnet3D.py:76-96) / PreActivationBasicBlock.forward (:41-57)."""
    residual = x
    out = F.relu(_bn(sd, x, p + ".bn1"))
    if cfg.block == "preact_bottleneck":
        out = _conv(sd, out, p + ".conv1", 1, 0)
        out = _conv(sd, F.relu(_bn(sd, out, p + ".bn2")), p + ".conv2", stride, 1)
        out = _conv(sd, F.relu(_bn(sd, out, p + ".bn3")), p + ".conv3", 1, 0)
    else:
        out = _conv(sd, out, p + ".conv1", stride, 1)
        out = _conv(sd, F.relu(_bn(sd, out, p + ".bn2")), p + ".conv2", 1, 1)
    if has_down:
        if cfg.shortcut == "A":
            residual = shortcut_a(x, planes * cfg.expansion, stride)
        else:
            residual = _bn(sd, _conv(sd, x, p + ".downsample.0", stride, 0), p + ".downsample.1")
    return out + residual


def _block(cfg, sd, x, p, planes, stride, has_down, nl):
    if cfg.block.startswith("preact"):
        return _preact_block(cfg, sd, x, p, planes, stride, has_down)
    residual = x
    if cfg.block == "resnext":       # ResNeXtBottleneck.forward, resnext3D.py:101-121
        out = F.relu(_bn(sd, _conv(sd, x, p + ".conv1", 1, 0), p + ".bn1"))
        out = F.conv3d(out, sd[p + ".conv2.weight"], None, _t3(stride), (1, 1, 1), 1, cfg.cardinality)
        out = F.relu(_bn(sd, out, p + ".bn2"))
        out = _bn(sd, _conv(sd, out, p + ".conv3", 1, 0), p + ".bn3")
    elif cfg.block in ("bottleneck", "wide"):      # WideBottleneck.forward (wideresnet3D.py:86-106) has the same flow
        out = F.relu(_bn(sd, _any_conv(cfg, sd, x, p + ".conv1", 1, 0), p + ".bn1"))
        out = F.relu(_bn(sd, _any_conv(cfg, sd, out, p + ".conv2", stride, 1), p + ".bn2"))
        out = _bn(sd, _any_conv(cfg, sd, out, p + ".conv3", 1, 0), p + ".bn3")
    else:
        out = F.relu(_bn(sd, _any_conv(cfg, sd, x, p + ".conv1", stride, 1), p + ".bn1"))
        out = _bn(sd, _any_conv(cfg, sd, out, p + ".conv2", 1, 1), p + ".bn2")
    if has_down:
        if cfg.shortcut == "A":
            residual = shortcut_a(x, planes * cfg.expansion, stride)
        else:
            residual = _bn(sd, _any_conv(cfg, sd, x, p + ".downsample.0", stride, 0),
                           p + ".downsample.1")
    out = F.relu(out + residual)
    if nl:
        out = nonlocal_block(sd, out, p + ".nonlocalblock", cfg.nl_mode)
    return out


def nl_flags(blocks, nonlocal_blocks):
    """Which blocks of a stage carry an NL block (nonlocalnet.py:474-479)."""
    freq = blocks // nonlocal_blocks if nonlocal_blocks != 0 else -1
    return [(i % freq == 0 and freq > 0) for i in range(blocks)]


def features(cfg: ArchCfg, sd, x):
    """`model.features(x)`: stem + maxpool + layer1..4 (torchvision_models.py:448-458)."""
    if cfg.dims == 2:
        x = F.relu(_bn(sd, _conv(sd, x, "conv1", 2, 3, 2), "bn1"))
        x = F.max_pool2d(x, 3, 2, 1)
    else:
        x = F.relu(_bn(sd, _any_conv(cfg, sd, x, "conv1", (1, 2, 2), (3, 3, 3)), "bn1"))
        x = F.max_pool3d(x, 3, 2, 1)
    inplanes = 64
    for li, (planes, nblocks) in enumerate(zip(cfg.widths, cfg.layers)):
        stride = 1 if li == 0 else 2
        flags = nl_flags(nblocks, cfg.nonlocal_layers[li]) if cfg.nonlocal_layers else [False] * nblocks
        for bi in range(nblocks):
            first = bi == 0
            has_down = first and (stride != 1 or inplanes != planes * cfg.expansion)
            x = _block(cfg, sd, x, "layer%d.%d" % (li + 1, bi), planes,
                       stride if first else 1, has_down, flags[bi])
            if first:
                inplanes = planes * cfg.expansion
    return x


def logits(cfg: ArchCfg, sd, feat):
    """`model.logits(features)`: global average pool + flatten + linear (:460-464)."""
    x = F.adaptive_avg_pool2d(feat, 1) if cfg.dims == 2 else F.adaptive_avg_pool3d(feat, 1)
    x = x.view(x.size(0), -1)
    return F.linear(x, sd[cfg.head + ".weight"], sd[cfg.head + ".bias"])


def forward(cfg: ArchCfg, sd, x):
    with torch.no_grad():
        return logits(cfg, sd, features(cfg, sd, x))


# --------------------------------------------------------------------------------------------
# pre-processing edge: the tensor half of TransformImage (transforms/utils.py:72-75)
# --------------------------------------------------------------------------------------------
def transform_frames(frames, mean, std, input_space="RGB", input_range=(0, 1)):
    """uint8 frames [N,T,H,W,C] -> fp32 [N,C,T,H,W], frame by frame as the reference composes it:
    torchvision ToTensor (`.to(float32).div(255)`, HWC -> CHW), ToSpaceBGR (utils.py:9-20),
    ToRange255 (:23-31), torchvision Normalize (`.sub_(mean).div_(std)`).  torchvision is an unpinned
    third-party dependency absent from this image; these are its documented semantics."""
    t = frames.permute(0, 4, 1, 2, 3).to(torch.float32).div(255)
    if input_space == "BGR":
        t = t[:, [2, 1, 0]].contiguous()
    if max(input_range) == 255:
        t = t.mul(255)
    m = torch.tensor(mean, dtype=torch.float32).view(1, -1, 1, 1, 1)
    s = torch.tensor(std, dtype=torch.float32).view(1, -1, 1, 1, 1)
    return t.sub(m).div(s)


# --------------------------------------------------------------------------------------------
# SlowFast (slowfast.py)
# --------------------------------------------------------------------------------------------
def _sf_block(sd, x, p, block, stride):
    """slowfast.Bottleneck.forward (:81-99) / BasicBlock.forward (:37-53).  The head conv's shape
    (1x1x1 | (3,1,1) | (1,3,3)) is read off the stored filter."""
    kt, kh, _ = sd[p + ".conv1.weight"].shape[2:]
    if block == "bottleneck":
        out = F.relu(_bn(sd, _conv(sd, x, p + ".conv1", 1, (kt // 2, 0, 0)), p + ".bn1"))
        out = F.relu(_bn(sd, _conv(sd, out, p + ".conv2", (1, stride, stride), (0, 1, 1)), p + ".bn2"))
        out = _bn(sd, _conv(sd, out, p + ".conv3", 1, 0), p + ".bn3")
    else:
        if kh == 3:     # head_conv == 1: (1,3,3) carrying the stride (:14-18)
            out = _conv(sd, x, p + ".conv1", (1, stride, stride), (0, 1, 1))
        else:           # head_conv == 3: (3,1,1), unstrided (:20-23)
            out = _conv(sd, x, p + ".conv1", 1, (1, 0, 0))
        out = F.relu(_bn(sd, out, p + ".bn1"))
        out = _bn(sd, _conv(sd, out, p + ".conv2", (1, stride, stride), (0, 1, 1)), p + ".bn2")   # conv2 has a bias
    residual = x
    if (p + ".downsample.0.weight") in sd:
        residual = _bn(sd, _conv(sd, x, p + ".downsample.0", (1, stride, stride), 0), p + ".downsample.1")
    return F.relu(out + residual)


def _sf_stage(sd, x, p, block, nblocks, stride):
    for bi in range(nblocks):
        x = _sf_block(sd, x, "%s.%d" % (p, bi), block, stride if bi == 0 else 1)
    return x


def _sf_stem(sd, x, p):
    kt = sd[p + "conv1.weight"].shape[2]
    x = F.relu(_bn(sd, _conv(sd, x, p + "conv1", (1, 2, 2), (kt // 2, 3, 3)), p + "bn1"))
    return F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))


def slowfast_forward(sd, x, block, layers, mode="sf", slow_stride=16, fast_stride=2):
    """slowfast.SlowFast.forward (:393-398) with Fast.forward (:280-299) and Slow.forward (:140-156);
    mode 's' / 'f': SlowOnly.forward (:227-241) / FastOnly.forward (:347-363).  Dropout: eval."""
    strides = (1, 2 if block == "bottleneck" else 1, 2, 2)
    names = ("res2", "res3", "res4", "res5")
    with torch.no_grad():
        if mode == "sf":
            f = _sf_stem(sd, x[:, :, ::fast_stride], "fast.")
            lateral = [F.conv3d(f, sd["fast.lateral_p1.weight"], None, (8, 1, 1), (2, 0, 0))]
            for name, n, st in zip(names, layers, strides):
                f = _sf_stage(sd, f, "fast." + name, block, n, st)
                if name != "res5":
                    lateral.append(F.conv3d(f, sd["fast.lateral_%s.weight" % name], None, (8, 1, 1), (2, 0, 0)))
            fast = F.adaptive_avg_pool3d(f, 1).view(-1, f.size(1))
            s = _sf_stem(sd, x[:, :, ::slow_stride], "slow.")
            for i, (name, n, st) in enumerate(zip(names, layers, strides)):
                s = torch.cat([s, lateral[i]], dim=1)
                s = _sf_stage(sd, s, "slow." + name, block, n, st)
            slow = F.adaptive_avg_pool3d(s, 1).view(-1, s.size(1))
            return F.linear(torch.cat([slow, fast], dim=1), sd["last_linear.weight"])
        h = _sf_stem(sd, x[:, :, ::(slow_stride if mode == "s" else fast_stride)], "")
        for name, n, st in zip(names, layers, strides):
            h = _sf_stage(sd, h, name, block, n, st)
        h = F.adaptive_avg_pool3d(h, 1).view(-1, h.size(1))
        return F.linear(h, sd["last_linear.weight"], sd["last_linear.bias"])


# --------------------------------------------------------------------------------------------
# TRN relation heads
# --------------------------------------------------------------------------------------------
def relation(sd, x, p, num_inputs):
    """trn.Relation.forward: ReLU -> Linear -> ReLU -> Linear on concatenated frames (trn.py:39-56)."""
    out_features = sd[p + "relate.3.weight"].shape[0]
    h = x.contiguous().view(-1, num_inputs * x.size(-1))
    h = F.linear(F.relu(h), sd[p + "relate.1.weight"], sd[p + "relate.1.bias"])
    h = F.linear(F.relu(h), sd[p + "relate.3.weight"], sd[p + "relate.3.bias"])
    return h.view(x.size(0), -1, out_features)


def multiscale_relation(sd, x, num_input, num_relations=3, rng=np.random):
    """trn.MultiScaleRelation.forward (trn.py:101-110): consumes the numpy RNG exactly as the
    reference does (one `choice` per scale, in scale order)."""
    scales = list(range(num_input, 1, -1))
    outs = []
    for si, scale in enumerate(scales):
        combos = list(itertools.combinations(range(num_input), scale))
        take = min(num_relations, len(combos))
        for idx in rng.choice(len(combos), take, replace=False):
            sub = x[..., combos[idx], :]
            outs.append(relation(sd, sub, "relations.%d." % si, scale))
    out_features = outs[0].shape[-1]
    return torch.stack(outs).sum(0).view(x.size(0), -1, out_features)


def hierarchical_relation(sd, x, p, num_inputs):
    """trn.HierarchicalRelation.forward (trn.py:150-160) for the only depth at which the reference
    runs (depth 0: relation_size >= num_inputs, which is what TRN always constructs, trn.py:230):
    `stack([final_relation(input)]).mean(0)`."""
    if any(k.startswith(p + "relations.") for k in sd):
        raise RuntimeError("depth >= 1: the reference forward raises in torch.stack")
    x = x.view(-1, num_inputs, x.size(-1))
    return torch.stack([relation(sd, x, p + "final_relation.", num_inputs)]).mean(0)


def trn_features(backbone_cfg, sd, x, num_segments, consensus="HTRN", rng=np.random):
    """trn.TRN.features (trn.py:246-255): per-frame backbone (its last_linear is a Dropout: identity
    in eval) -> [B, 1, T, F] -> temporal relation -> squeeze()."""
    bsd = {k[len("base_model."):]: v for k, v in sd.items() if k.startswith("base_model.")}
    frames = x.reshape((-1, 3) + tuple(x.shape[-2:]))
    feat = features(backbone_cfg, bsd, frames)
    pooled = F.adaptive_avg_pool2d(feat, 1).view(feat.size(0), -1)
    rep = pooled.view(x.size(0), -1, num_segments, pooled.size(-1))
    tsd = {k[len("temporal_relation."):]: v for k, v in sd.items() if k.startswith("temporal_relation.")}
    if consensus == "TRN":
        out = relation(tsd, rep, "", num_segments)
    elif consensus == "HTRN":
        out = hierarchical_relation(tsd, rep, "", num_segments)
    elif consensus == "MSTRN":
        out = multiscale_relation(tsd, rep, num_segments, 3, rng)
    else:
        raise ValueError("Unrecognized temporal consensus.")
    return out.squeeze()


def trn_forward(backbone_cfg, sd, x, num_segments, consensus="HTRN", rng=np.random):
    """trn.TRN.forward (trn.py:257-263)."""
    with torch.no_grad():
        f = trn_features(backbone_cfg, sd, x, num_segments, consensus, rng)
        return F.linear(f, sd["last_linear.weight"], sd["last_linear.bias"])


# --------------------------------------------------------------------------------------------
# nominal work (GMAC) -- the figure roofline numbers are quoted against, SURVEY.md section 8d
# --------------------------------------------------------------------------------------------
def conv_macs(out_numel, cin, k):
    kt, kh, kw = _t3(k)
    return out_numel * cin * kt * kh * kw
