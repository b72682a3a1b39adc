"""Deterministic synthetic weights and inputs (there is no network for checkpoints/datasets).

The recipe is the calibrated one of SURVEY.md section 8(d): default-initialised BatchNorms are the
identity and the non-local block's output BN is all-zero (reference nonlocalnet.py:95-96), so a
parity check on default-init weights would be vacuous for BN folding and for the whole NL branch.
Every tensor is drawn from its own CPU generator seeded by (seed, crc32(key)), so the values
depend only on the state_dict key and shape -- not on module construction order -- and the same
function feeds the reference model (golden generation), the oracle and the HIP engine.
"""
import zlib
from collections import OrderedDict

import torch

# gamma of the BN that closes a residual branch is damped so that logits stay O(10) deep into
# the network (factor 0.8 -> max|logit| ~ 26 for resnet3d50).
LAST_BN_DAMP = 0.8
# the non-local branch re-injects a full-width signal after the block's ReLU; its output BN
# (`W.1`) is damped harder so NL networks stay in the same logit range (calibrated: ~10-30)
NL_BN_DAMP = 0.2


# I3D (Inception stacks, no residual paths): fan_in filters + mildly damped Unit3D BNs keep
# max|logit| ~ 16 at 16x224x224 with an fp32 noise floor of ~6e-6 (calibrated in round 1)
I3D_RECIPE = dict(unit_bn_damp=0.9, conv_fan="in")


# BigGAN-deep generator (pre-activation residual stack, no BN after the closing convs): fan_in filters,
# closing convs x0.2 -> pre-tanh std ~0.9, max ~4 at 256x256, fp32 noise floor ~4e-6
BIGGAN_RECIPE = dict(conv_fan="in", closing_conv_damp=0.2)


def _gen(seed, key):
    g = torch.Generator()
    g.manual_seed((int(seed) * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
    return g


def _is_closing_bn(prefix, keys):
    if prefix.endswith(".bn3") or prefix.endswith("downsample.1"):
        return True
    if prefix.endswith(".bn2"):
        parent = prefix.rsplit(".", 1)[0]
        return (parent + ".bn3.weight") not in keys
    return False


def synth_state_dict(template, seed=1234, last_bn_damp=LAST_BN_DAMP, nl_bn_damp=NL_BN_DAMP, inner_bn_damp=1.0,
                     unit_bn_damp=1.0, conv_fan="out", closing_conv_damp=1.0, nl_embed_damp=1.0):
    """template: mapping key -> tensor (only shape/dtype are used). Returns an OrderedDict of
    fresh CPU fp32 tensors with the same keys/shapes.

    The damping factors scale BN gammas so that logits of deep random-weight networks stay in the
    calibrated 10-30 range (SURVEY.md 8d "re-calibrate per model"): `last_bn_damp` for the BN that
    closes a residual branch, `nl_bn_damp` for the non-local block's output BN (`W.1`),
    `inner_bn_damp` for the BN inside a (2+1)D factored conv pair (`*.bn`), `unit_bn_damp` for the BN
    of an I3D Unit3D (`*.bn` next to `*.conv3d`).  `conv_fan`: 'out' = the reference's kaiming-normal
    fan_out (resnet3D.py:198); 'in' = fan_in, for Inception stacks whose 1x1x1 reductions (192 -> 16)
    would otherwise amplify 2*Cin/Cout per layer; `closing_conv_damp` scales the filters
    that close a BigGAN-deep residual branch (`conv4`, attention `o`), which has no BN after them;
    `nl_embed_damp` scales the non-local block's theta / phi embeddings (weights and biases): the softmax input
    theta^T phi grows with the SQUARE of the activation scale, and with kaiming filters it reaches 1e2 ... 1e4 in a
    random-weight network -- a hard arg-max whose winner flips under fp32 rounding (round 4: the reference's own CPU
    forward then differs from its fp64 self by 1e-2 relative).  damp^2 brings the affinities back to the O(1-30) range of
    a trained network, so the NL branch can be tested at FULL strength (`nl_bn_damp` = 1).  Fixtures record the values
    they were generated with."""
    keys = set(template.keys())
    out = OrderedDict()

    def embed_damp(key):
        # a float, or {key prefix: factor} (activations grow with depth, and the affinities with their square)
        if isinstance(nl_embed_damp, dict):
            return next((float(v) for k, v in nl_embed_damp.items() if key.startswith(k)), 1.0)
        return float(nl_embed_damp)

    for key, ref in template.items():
        shape = tuple(ref.shape)
        g = _gen(seed, key)
        prefix, _, leaf = key.rpartition(".")
        is_bn = (prefix + ".running_mean") in keys
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros(shape, dtype=torch.long)
        elif is_bn and leaf == "weight":
            v = torch.rand(shape, generator=g) + 0.5
            if prefix.endswith(".W.1"):
                v = v * nl_bn_damp
            elif prefix.endswith(".bn") and (prefix[:-3] + ".spatial_conv.weight") in keys:
                v = v * inner_bn_damp
            elif prefix.endswith(".bn") and (prefix[:-3] + ".conv3d.weight") in keys:
                v = v * unit_bn_damp
            elif _is_closing_bn(prefix, keys):
                v = v * last_bn_damp
            out[key] = v
        elif is_bn and leaf == "bias":
            out[key] = torch.randn(shape, generator=g) * 0.1
        elif leaf in ("running_mean", "stored_mean"):
            out[key] = torch.randn(shape, generator=g) * 0.1
        elif leaf in ("running_var", "stored_var"):
            out[key] = torch.rand(shape, generator=g) + 0.5
        elif leaf == "gain":                                # BigGAN's plain output BN: gamma
            out[key] = torch.rand(shape, generator=g) + 0.5
        elif leaf == "gamma":                               # attention mixing scalar (init 0 upstream)
            out[key] = torch.full(shape, 0.5)
        elif leaf == "weight" and len(shape) >= 3:          # conv: kaiming-normal, fan_out
            fan = shape[0] if conv_fan == "out" else shape[1]
            for k in shape[2:]:
                fan *= k
            out[key] = torch.randn(shape, generator=g) * (2.0 / fan) ** 0.5
            if prefix.endswith(".conv4") or prefix.endswith(".o"):      # closes a BigGAN residual / attention branch
                out[key] = out[key] * closing_conv_damp
            if ".nonlocalblock." in key and (prefix.endswith(".theta") or prefix.endswith(".phi") or prefix.endswith(".phi.0")):
                out[key] = out[key] * embed_damp(key)
        elif leaf == "weight" and shape == (1, 3) and prefix.endswith(".linear") and (prefix[:-7] + ".weight") in keys:
            # MultiViewConv's view-mixing Linear(3, 1) (multiview.py:50): positive weights with sum of squares ~ 1, so the
            # three-view sum keeps the signal scale of a plain conv (default Linear init would shrink it 0.58x per layer)
            out[key] = torch.rand(shape, generator=g) * 0.6 + 0.3
        elif leaf == "weight" and len(shape) == 2:          # linear
            bound = 1.0 / shape[1] ** 0.5
            out[key] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        elif leaf == "bias":                                # conv / linear bias
            out[key] = (torch.rand(shape, generator=g) * 2 - 1) * 0.05
            if ".nonlocalblock." in key and (prefix.endswith(".theta") or prefix.endswith(".phi") or prefix.endswith(".phi.0")):
                out[key] = out[key] * embed_damp(key)
        else:
            raise KeyError("synth_state_dict: unclassified key %r" % key)
    return out


def synth_clips(batch, frames, size, seed=99, channels=3):
    """Synthetic normalised video clips, NCDHW fp32, from a seeded CPU generator."""
    g = torch.Generator()
    g.manual_seed(int(seed))
    return torch.randn(batch, channels, frames, size, size, generator=g)
