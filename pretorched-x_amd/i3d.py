"""I3D (Inception-v1 inflated to 3-D; Carreira & Zisserman, "Quo Vadis, Action Recognition?", 2017)
behind the HIP engine -- BASELINE.json config 4.

The mounted reference snapshot has NO I3D source: this module follows
the published architecture with the layer names of the widely used PyTorch port of DeepMind's
kinetics-i3d (`InceptionI3d`: Conv3d_1a_7x7 ... Mixed_5c, logits; Unit3D = conv (no bias) +
BatchNorm(eps 1e-3) + ReLU with TF-"SAME" padding), so such checkpoints load by key.  **Parity is
unpinned by the reference**: the checker is the builder-written CPU module in oracle/i3d_standin.py.

Only parameters live here; `engine.Plan._build_i3d` compiles the forward pass: every Unit3D is one
implicit-GEMM launch with BN/ReLU folded, "SAME" padding is front-pad geometry (no F.pad copy), the
four branches of an Inception module write their channel slices of the module output directly
(no torch.cat), the stem is the kW-folded small-Cin path.
"""
import torch.nn as nn

from .engine import EngineOwner
from .zoo import Arch, Bag


def _unit(cin, cout, k=(1, 1, 1), stride=(1, 1, 1), bn=True, bias=False):
    u = Bag()
    u.conv3d = nn.Conv3d(cin, cout, k, stride, padding=0, bias=bias)
    u.conv3d.tf_same = True            # padding is computed per input size (TF "SAME"), not stored
    if bn:
        u.bn = nn.BatchNorm3d(cout, eps=0.001, momentum=0.01)
    u.has_bn = bn
    return u


def _inception(cin, o):
    m = Bag()
    m.b0 = _unit(cin, o[0])
    m.b1a = _unit(cin, o[1])
    m.b1b = _unit(o[1], o[2], (3, 3, 3))
    m.b2a = _unit(cin, o[3])
    m.b2b = _unit(o[3], o[4], (3, 3, 3))
    m.b3b = _unit(cin, o[5])           # after a 3x3x3 stride-1 SAME max pool (b3a, parameterless)
    m.out_channels = o[0] + o[2] + o[4] + o[5]
    m.splits = (o[0], o[2], o[4], o[5])
    return m


# name -> (input channels, branch widths), in execution order; "pool*" entries are SAME max pools
_LAYOUT = (
    ("Mixed_3b", 192, (64, 96, 128, 16, 32, 32)),
    ("Mixed_3c", 256, (128, 128, 192, 32, 96, 64)),
    ("pool4a", (3, 3, 3), (2, 2, 2)),
    ("Mixed_4b", 480, (192, 96, 208, 16, 48, 64)),
    ("Mixed_4c", 512, (160, 112, 224, 24, 64, 64)),
    ("Mixed_4d", 512, (128, 128, 256, 24, 64, 64)),
    ("Mixed_4e", 512, (112, 144, 288, 32, 64, 64)),
    ("Mixed_4f", 528, (256, 160, 320, 32, 128, 128)),
    ("pool5a", (2, 2, 2), (2, 2, 2)),
    ("Mixed_5b", 832, (256, 160, 320, 32, 128, 128)),
    ("Mixed_5c", 832, (384, 192, 384, 48, 128, 128)),
)


class InceptionI3d(EngineOwner, nn.Module):
    """[B,3,T,224,224] -> [B,num_classes]: per-frame logits averaged over the remaining time steps (the
    original model's `reduce_mean(logits, axis=1)`).  `features` returns the Mixed_5c map."""
    plan_kind = "i3d"
    layout = _LAYOUT

    def __init__(self, num_classes=400, in_channels=3, dropout_keep_prob=0.5):
        super().__init__()
        self.num_classes = num_classes
        self.arch = Arch("inception", (), "B")
        self.Conv3d_1a_7x7 = _unit(in_channels, 64, (7, 7, 7), (2, 2, 2))
        self.Conv3d_2b_1x1 = _unit(64, 64)
        self.Conv3d_2c_3x3 = _unit(64, 192, (3, 3, 3))
        for entry in _LAYOUT:
            if entry[0].startswith("Mixed"):
                setattr(self, entry[0], _inception(entry[1], entry[2]))
        self.dropout = nn.Dropout(dropout_keep_prob)
        self.logits = _unit(1024, num_classes, bn=False, bias=True)
        self.eval()
        self._init_engine()

    @property
    def head_module(self):
        return self.logits.conv3d

    def replace_logits(self, num_classes):
        """New classifier for fine-tuned checkpoints (the port's `replace_logits`)."""
        self.num_classes = num_classes
        old = self.logits.conv3d
        self.logits = _unit(1024, num_classes, bn=False, bias=True)
        self.logits.to(old.weight.device)
        self._engine.invalidate()

    def features(self, input):
        return self._engine.features(self, input)

    def forward(self, input):
        return self._engine.forward(self, input)

    def forward_frames(self, frames, opts):
        return self._engine.forward_frames(self, frames, opts)


def i3d(num_classes=400, pretrained=None):
    """BASELINE.json config 4 (InceptionV1-3D, Kinetics-400).  No checkpoint URL is published by the
    reference snapshot; `pretrained` must be None (load a converted state_dict yourself)."""
    if pretrained is not None:
        raise ValueError("no pretrained I3D weights are published for this package (no network, no reference URL)")
    return InceptionI3d(num_classes)
