"""torch.nn execution of the zoo's parameter trees -- ONLY for what the HIP engine cannot serve by definition.

The HIP engine is forward-only inference on ROCm tensors.  The reference's modules also run in `train()` mode
(batch-statistics BatchNorm, autograd: `/root/reference/pretorched/models/torchvision_models.py:448-469` are plain
nn.Module calls) and on CPU tensors (BASELINE.json config 1 is literally the CPU path).  SURVEY.md section 8(b)
"Tensors": "the engine must fall back to the PyTorch path when `self.training` or when grad is required".  This
module is that path: the zoo's own nn.Conv / nn.BatchNorm / nn.Linear children are CALLED, in the reference's op
order, so autograd, BN running-statistic updates and CPU execution behave as upstream.

Selection rule (`wanted`): training mode, or an input that requires grad, or CPU input with CPU parameters, or --
opt-in, `model.engine().autograd = True` -- grad mode on with trainable parameters (eval-mode fine-tuning).
Never in eval mode on ROCm tensors: that is always the HIP path, which raises when libptx_amd.so is missing or a
launch fails -- this module is not a fallback for it, and nothing under oracle/ is used here.
`PTX_EAGER=0` turns it off (those calls then raise PtxError as in round 1).
`calls` counts eager forwards so the GPU test-suite can assert it never ran.

Reference op order followed (file:line under /root/reference/pretorched/models):
  resnet3D.py:93-106 (BasicBlock), :125-143 (Bottleneck), :65-74 (shortcut A), :203-218 (forward);
  pre_act_resnet3D.py:41-57, :76-96; resnext3D.py:101-121; wideresnet3D.py:86-106; r2plus1d.py:85-88;
  nonlocalnet.py:139-243 (the four NL modes), :402-420 (NonLocalBottleneck); trn.py:39-56, :95-113.
"""
import os

import torch
import torch.nn.functional as F

from ._lib import PtxError

calls = 0


def enabled():
    return os.environ.get("PTX_EAGER", "1") != "0"


def wanted(model, x):
    """True when this call is outside the HIP engine's contract and the torch.nn path must serve it."""
    if not enabled() or not isinstance(x, torch.Tensor):
        return False
    if model.training or (torch.is_grad_enabled() and x.requires_grad):
        return True
    # Opt-in: eval-mode fine-tuning (frozen BN statistics, trainable trunk).  The reference returns a differentiable
    # output whenever grad mode is on and a parameter requires grad; the HIP engine is forward-only, so by default an
    # eval-mode call on ROCm tensors returns logits WITHOUT a grad_fn (every nn.Parameter requires grad by default --
    # routing on that alone would send plain inference to torch.nn).  `model.engine().autograd = True` (or
    # PTX_AUTOGRAD=1) asks for the reference behaviour: such calls then run the torch.nn children.
    eng = getattr(model, "_engine", None)
    if eng is not None and getattr(eng, "autograd", False) and torch.is_grad_enabled():
        if any(p.requires_grad for p in model.parameters()):
            return True
    if not x.is_cuda:
        from .engine import _first_weight
        return not _first_weight(model).is_cuda        # CPU model + CPU input; a device mismatch still raises
    return False


def _count():
    global calls
    calls += 1


# ---------------------------------------------------------------------------------------------
def conv(m, x):
    """nn.Conv{2,3}d, or a (2+1)D pair: spatial conv -> BN -> ReLU -> temporal conv (r2plus1d.py:85-88)."""
    if hasattr(m, "spatial_conv"):
        return m.temporal_conv(F.relu(m.bn(m.spatial_conv(x))))
    return m(x)


def shortcut_a(x, planes, stride):
    """resnet3D.py:65-74: strided subsample + zero channels (allocated on x's device / dtype)."""
    out = F.avg_pool3d(x, kernel_size=1, stride=stride)
    pad = torch.zeros(out.size(0), planes - out.size(1), *out.shape[2:], dtype=out.dtype, device=out.device)
    return torch.cat([out, pad], dim=1)


def nonlocal_block(nl, x):
    """_NonLocalBlockND.forward for dimension 3 (nonlocalnet.py:139-243); `nl` is a zoo Bag or NonLocalBlock3D."""
    mode = getattr(nl, "mode", "embedded_gaussian")
    b, c = x.shape[:2]
    sub = bool(getattr(nl, "sub_sample", False))

    g_x = nl.g(x)                         # Sequential(conv, MaxPool3d(2)) when sub-sampling
    ci = g_x.shape[1]
    g_x = g_x.reshape(b, ci, -1).permute(0, 2, 1)
    if mode == "gaussian":
        theta_x = x.reshape(b, c, -1).permute(0, 2, 1)
        phi_x = (nl.phi(x) if sub else x).reshape(b, c, -1)
        f = F.softmax(torch.matmul(theta_x, phi_x), dim=-1)
    else:
        theta = nl.theta(x)
        phi = nl.phi(x)
        if mode == "concatenation":
            th = theta.reshape(b, ci, -1, 1)
            ph = phi.reshape(b, ci, 1, -1)
            h, w = th.size(2), ph.size(3)
            f = nl.concat_project(torch.cat([th.repeat(1, 1, 1, w), ph.repeat(1, 1, h, 1)], dim=1))
            f = f.reshape(b, h, w)
            f = f / f.size(-1)
        else:
            f = torch.matmul(theta.reshape(b, ci, -1).permute(0, 2, 1), phi.reshape(b, ci, -1))
            f = F.softmax(f, dim=-1) if mode == "embedded_gaussian" else f / f.size(-1)
    y = torch.matmul(f, g_x).permute(0, 2, 1).contiguous().reshape(b, ci, *x.shape[2:])
    W = nl.W
    if isinstance(W, torch.nn.ModuleList):            # zoo Bag: [conv, bn]
        w_y = W[1](W[0](y))
    else:
        w_y = W(y)
    return w_y + x


def _residual(arch, blk, x):
    if not blk.has_shortcut:
        return x
    if arch.shortcut == "B":
        return blk.downsample[1](conv(blk.downsample[0], x))
    planes = blk.bn3.num_features if hasattr(blk, "bn3") and not arch.block.startswith("preact") else None
    if planes is None:                                  # basic / pre-activation blocks: width of the block output
        last = blk.conv3 if hasattr(blk, "conv3") else blk.conv2
        planes = (last.temporal_conv if hasattr(last, "spatial_conv") else last).out_channels
    return shortcut_a(x, planes, blk.stride)


def block(arch, blk, x):
    if arch.block.startswith("preact"):                 # pre_act_resnet3D.py:41-57 / :76-96
        out = F.relu(blk.bn1(x))
        out = conv(blk.conv1, out)
        out = conv(blk.conv2, F.relu(blk.bn2(out)))
        if arch.block == "preact_bottleneck":
            out = conv(blk.conv3, F.relu(blk.bn3(out)))
        return out + _residual(arch, blk, x)
    out = F.relu(blk.bn1(conv(blk.conv1, x)))
    if arch.block in ("bottleneck", "resnext", "wide"):
        out = F.relu(blk.bn2(conv(blk.conv2, out)))
        out = blk.bn3(conv(blk.conv3, out))
    else:
        out = blk.bn2(conv(blk.conv2, out))
    out = F.relu(out + _residual(arch, blk, x))
    if blk.has_nl:
        out = nonlocal_block(blk.nonlocalblock, out)
    return out


def resnet_features(model, x):
    _count()
    arch = model.arch
    x = F.relu(model.bn1(conv(model.conv1, x)))
    x = model.maxpool(x)
    for li in range(4):
        for blk in getattr(model, "layer%d" % (li + 1)):
            x = block(arch, blk, x)
    return x


def resnet_logits(model, feats):
    x = model.avgpool(feats)
    x = x.view(x.size(0), -1)
    return model.head_module(x)


def resnet_forward(model, x):
    return resnet_logits(model, resnet_features(model, x))


def relation(rel, x):
    """Relation.forward (trn.py:39-56)."""
    _count()
    out = rel.relate(x.contiguous().view(-1, rel.num_inputs * rel.in_features))
    return out.view(x.size(0), -1, rel.out_features)
