"""BigGAN-deep generator (Brock, Donahue, Simonyan: "Large Scale GAN Training for High Fidelity Natural
Image Synthesis", ICLR 2019, appendix B "BigGAN-deep") behind the HIP engine -- BASELINE.json config 5.

The mounted reference snapshot has NO BigGAN source.  This module
follows the published architecture with the parameter names of the authors' PyTorch release
(`BigGANdeep.Generator`: shared, linear, blocks.{stage}.{j}.{conv1..4, bn1..4.{gain,bias,stored_mean,
stored_var}}, attention {theta,phi,g,o,gamma}, output_layer.{0,2}).  **Parity is unpinned by the
reference**: the checker is the builder-written CPU module in oracle/biggan_standin.py.

Spectral normalisation is a weight reparametrisation (W / sigma(W)); checkpoints are expected with it
folded in (the `u*` / `sv*` power-iteration buffers are ignored on load).

Round-1 arithmetic: fp32 MFMA through the same implicit-GEMM kernel as the video nets (config 5 names
fp16 MFMA; an fp16 tile family is future work).  What is fused: class-conditional BN + ReLU + nearest
upsample run as one HBM pass (`ptx_affine_act_upsample`) over a table of per-sample scale/shift values
folded from ONE linear over all 50 cBN layers (`ptx_linear_fwd` + `ptx_cbn_fold`); the skip connection
`upsample(x[:, :Cout])` is a gather in the closing conv's epilogue (no upsampled copy of x exists).
"""
import torch
import torch.nn as nn

from .engine import EngineOwner
from .zoo import Arch, Bag


def _ccbn(channels, cond_dim):
    """layers.ccbn: gain/bias are bias-free Linears of the conditioning vector; stored statistics."""
    bn = Bag()
    bn.gain = nn.Linear(cond_dim, channels, bias=False)
    bn.bias = nn.Linear(cond_dim, channels, bias=False)
    bn.register_buffer("stored_mean", torch.zeros(channels))
    bn.register_buffer("stored_var", torch.ones(channels))
    bn.channels = channels
    return bn


def _gblock(cin, cout, cond_dim, upsample, ratio=4):
    b = Bag()
    hid = cin // ratio
    b.conv1 = nn.Conv2d(cin, hid, 1)
    b.conv2 = nn.Conv2d(hid, hid, 3, padding=1)
    b.conv3 = nn.Conv2d(hid, hid, 3, padding=1)
    b.conv4 = nn.Conv2d(hid, cout, 1)
    b.bn1, b.bn2, b.bn3, b.bn4 = _ccbn(cin, cond_dim), _ccbn(hid, cond_dim), _ccbn(hid, cond_dim), _ccbn(hid, cond_dim)
    b.in_channels, b.out_channels, b.hidden, b.upsample, b.kind = cin, cout, hid, upsample, "gblock"
    return b


def _attention(ch):
    a = Bag()
    a.theta = nn.Conv2d(ch, ch // 8, 1, bias=False)
    a.phi = nn.Conv2d(ch, ch // 8, 1, bias=False)
    a.g = nn.Conv2d(ch, ch // 2, 1, bias=False)
    a.o = nn.Conv2d(ch // 2, ch, 1, bias=False)
    a.gamma = nn.Parameter(torch.tensor(0.0))
    a.ch, a.kind = ch, "attention"
    return a


# resolution -> (in multipliers, out multipliers, attention resolution)
_ARCH = {
    256: ((16, 16, 8, 8, 4, 2), (16, 8, 8, 4, 2, 1), 64),
    128: ((16, 16, 8, 4, 2), (16, 8, 4, 2, 1), 64),
    64: ((16, 16, 8, 4), (16, 8, 4, 2), 64),
    32: ((4, 4, 4), (4, 4, 4), None),
}


class BigGANDeepGenerator(EngineOwner, nn.Module):
    """forward(z [B,dim_z], y [B,shared_dim] = self.shared(labels)) -> images [B,3,R,R] in (-1, 1)."""
    plan_kind = "biggan"

    def __init__(self, resolution=256, ch=128, dim_z=128, shared_dim=128, n_classes=1000, depth=2, bottom_width=4,
                 bn_eps=1e-5, precision="fp32"):
        super().__init__()
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' or 'fp16'")
        # 'fp16': the cBN -> ReLU (-> upsample) passes emit halfs and every GBlock / output conv runs on fp16 MFMA
        # with fp32 accumulation, bias, skip connection and output (BASELINE.json config 5 names fp16 MFMA)
        self.precision = precision
        if resolution not in _ARCH:
            raise ValueError("resolution must be one of %s" % sorted(_ARCH))
        ins_, outs_, attn_ = _ARCH[resolution]
        if depth != 2:
            raise ValueError("BigGAN-deep GBlocks come in pairs (depth = 2): the second block of a pair upsamples; got depth=%r" % (depth,))
        if dim_z % 4 or shared_dim % 4 or dim_z <= 0 or shared_dim <= 0:
            raise ValueError("dim_z and shared_dim must be positive multiples of 4 (16-byte rows of the conditioning vector); "
                             "got dim_z=%r shared_dim=%r" % (dim_z, shared_dim))
        if ch <= 0 or (min(outs_) * ch) % 8:
            raise ValueError("ch * %d (the narrowest stage) must be a multiple of 8 channels; got ch=%r" % (min(outs_), ch))
        if attn_ is not None:
            att_ch = [co * ch for i, co in enumerate(outs_) if bottom_width * 2 ** (i + 1) == attn_]
            if any(c % 32 for c in att_ch):
                raise ValueError("the self-attention stage needs a channel count that is a multiple of 32 (theta / phi use "
                                 "ch/8 channels in 4-channel groups); got %s from ch=%r" % (att_ch, ch))
        self.resolution, self.ch, self.dim_z, self.shared_dim = resolution, ch, dim_z, shared_dim
        self.n_classes, self.depth, self.bottom_width, self.bn_eps = n_classes, depth, bottom_width, bn_eps
        self.arch = Arch("gblock", (), "B", dims=2)
        ins, outs, attn_res = _ARCH[resolution]
        cond = dim_z + shared_dim                      # hier: every cBN sees cat([shared(y), z])
        self.cond_dim = cond
        self.shared = nn.Embedding(n_classes, shared_dim)
        self.linear = nn.Linear(cond, ins[0] * ch * bottom_width ** 2)
        blocks = []
        res = bottom_width
        for i, (ci, co) in enumerate(zip(ins, outs)):
            stage = [_gblock(ci * ch, (ci if d == 0 else co) * ch, cond, upsample=(d == depth - 1)) for d in range(depth)]
            res *= 2
            if attn_res is not None and res == attn_res:
                stage.append(_attention(co * ch))
            blocks.append(nn.ModuleList(stage))
        self.blocks = nn.ModuleList(blocks)
        out_bn = Bag()                                 # layers.bn: plain BN with gain/bias parameters
        out_bn.gain = nn.Parameter(torch.ones(outs[-1] * ch))
        out_bn.bias = nn.Parameter(torch.zeros(outs[-1] * ch))
        out_bn.register_buffer("stored_mean", torch.zeros(outs[-1] * ch))
        out_bn.register_buffer("stored_var", torch.ones(outs[-1] * ch))
        out_bn.channels = outs[-1] * ch
        self.output_layer = nn.ModuleList([out_bn, nn.ReLU(), nn.Conv2d(outs[-1] * ch, 3, 3, padding=1)])
        self.eval()
        self._init_engine()

    def load_state_dict(self, state_dict, strict=True, **kw):
        # spectral-norm power-iteration buffers of the original release are not parameters here
        sd = {k: v for k, v in state_dict.items() if not (k.rsplit(".", 1)[-1].startswith(("u", "sv"))
                                                          and k.rsplit(".", 1)[-1][1:].lstrip("v").isdigit())}
        return super().load_state_dict(self._from_release_layout(sd), strict=strict, **kw)

    def _from_release_layout(self, sd):
        """The authors' BigGANdeep.Generator nests one ModuleList per GBlock -- keys `blocks.{stage*depth + d}.0.*`, the
        attention block riding as `blocks.{j}.1.*` on the last GBlock of its stage -- where this class nests one list
        per STAGE (`blocks.{stage}.{d}.*`, attention at `blocks.{stage}.{depth}.*`).  A state_dict in the release
        layout is recognised by its block indices running past the number of stages and is re-keyed."""
        import re
        idx = [int(m.group(1)) for m in (re.match(r"blocks\.(\d+)\.", k) for k in sd) if m]
        if not idx or max(idx) < len(self.blocks):
            return sd
        out = {}
        for k, v in sd.items():
            m = re.match(r"blocks\.(\d+)\.(\d+)\.(.*)", k)
            if m:
                stage, d = divmod(int(m.group(1)), self.depth)
                k = "blocks.%d.%d.%s" % (stage, d if int(m.group(2)) == 0 else self.depth, m.group(3))
            out[k] = v
        return out

    def forward(self, z, y):
        return self._engine.generate(self, z, y)


def biggan_deep(resolution=256, pretrained=None, **kwargs):
    """BASELINE.json config 5 (BigGAN-deep-256 generator).  No checkpoint is published by the reference
    snapshot; `pretrained` must be None."""
    if pretrained is not None:
        raise ValueError("no pretrained BigGAN weights are published for this package (no network, no reference URL)")
    return BigGANDeepGenerator(resolution, **kwargs)
