"""TEST INFRASTRUCTURE ONLY -- CPU restatement of I3D (Inception-v1 3-D) for BASELINE.json config 4.

**Parity unpinned**: the mounted reference snapshot contains no I3D source, so
this is NOT a restatement of reference code.  It follows the published architecture (Carreira &
Zisserman 2017, Inception-v1 inflated) as implemented by the common PyTorch port of DeepMind's
kinetics-i3d (`InceptionI3d`), whose semantics are:
  * Unit3D = Conv3d(padding=0, bias=False) on an explicitly zero-padded input (TF "SAME":
    pad_total = max(k - stride, 0) if in % stride == 0 else max(k - in % stride, 0), front = total // 2)
    -> BatchNorm3d(eps=1e-3) -> ReLU;
  * MaxPool3dSamePadding = the same zero F.pad followed by MaxPool3d(padding=0);
  * InceptionModule = cat([b0(x), b1b(b1a(x)), b2b(b2a(x)), b3b(maxpool3x3x3_s1(x))], dim=1);
  * head = AvgPool3d([2,7,7], stride 1) -> Dropout -> Unit3D(1024 -> classes, 1x1x1, bias, no BN, no
    activation) -> squeeze(3).squeeze(3) -> [B, classes, T']; the clip prediction is the mean over T'
    (the original model's reduce_mean over time).
Functional, driven by a state_dict with that port's key names (the same ones pretorched_x_amd.i3d uses).
"""
import torch
import torch.nn.functional as F

LAYOUT = ("Mixed_3b", "Mixed_3c", ("pool", (3, 3, 3), (2, 2, 2)), "Mixed_4b", "Mixed_4c", "Mixed_4d", "Mixed_4e",
          "Mixed_4f", ("pool", (2, 2, 2), (2, 2, 2)), "Mixed_5b", "Mixed_5c")


def _pad_same(x, k, s):
    pads = []
    for dim, kk, ss in zip(x.shape[2:], k, s):
        total = max(kk - ss, 0) if dim % ss == 0 else max(kk - dim % ss, 0)
        pads.append((total // 2, total - total // 2))
    (tf, tb), (hf, hb), (wf, wb) = pads
    return F.pad(x, (wf, wb, hf, hb, tf, tb))


def unit3d(sd, x, p, stride=(1, 1, 1), bn=True, relu=True):
    w = sd[p + ".conv3d.weight"]
    x = F.conv3d(_pad_same(x, w.shape[2:], stride), w, sd.get(p + ".conv3d.bias"), stride)
    if bn:
        x = F.batch_norm(x, sd[p + ".bn.running_mean"], sd[p + ".bn.running_var"], sd[p + ".bn.weight"],
                         sd[p + ".bn.bias"], False, 0.01, 0.001)
    return F.relu(x) if relu else x


def maxpool_same(x, k, s):
    return F.max_pool3d(_pad_same(x, k, s), k, s)


def inception(sd, x, p):
    b0 = unit3d(sd, x, p + ".b0")
    b1 = unit3d(sd, unit3d(sd, x, p + ".b1a"), p + ".b1b")
    b2 = unit3d(sd, unit3d(sd, x, p + ".b2a"), p + ".b2b")
    b3 = unit3d(sd, maxpool_same(x, (3, 3, 3), (1, 1, 1)), p + ".b3b")
    return torch.cat([b0, b1, b2, b3], dim=1)


def features(sd, x):
    """-> Mixed_5c map [B,1024,T/8,H/32,W/32]."""
    with torch.no_grad():
        x = unit3d(sd, x, "Conv3d_1a_7x7", (2, 2, 2))
        x = maxpool_same(x, (1, 3, 3), (1, 2, 2))
        x = unit3d(sd, x, "Conv3d_2b_1x1")
        x = unit3d(sd, x, "Conv3d_2c_3x3")
        x = maxpool_same(x, (1, 3, 3), (1, 2, 2))
        for entry in LAYOUT:
            x = maxpool_same(x, entry[1], entry[2]) if isinstance(entry, tuple) else inception(sd, x, entry)
        return x


def per_frame_logits(sd, x):
    """-> [B, classes, T'] (the port's forward output)."""
    with torch.no_grad():
        f = F.avg_pool3d(features(sd, x), (2, 7, 7), (1, 1, 1))
        return unit3d(sd, f, "logits", bn=False, relu=False).squeeze(3).squeeze(3)


def forward(sd, x):
    """-> [B, classes]: mean of the per-frame logits over time."""
    return per_frame_logits(sd, x).mean(2)
