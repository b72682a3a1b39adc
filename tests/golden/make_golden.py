"""Generate tests/golden/*.npz by running the REAL reference (/root/reference, via oracle.ref_shim)
on CPU with the deterministic synthetic weights/inputs of pretorched_x_amd.testing.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [case ...]

With case names only those fixtures are regenerated (state_keys.json is merged, not rewritten);
`trn` selects the TRN head + wrapper fixtures.

The reference tree does not exist on the GPU box, so these small fixtures (logits, small feature
maps, state_dict key lists) are committed; the GPU parity tests compare the HIP engine against
them and, in the same run, against the oracle restatement (oracle/functional.py).

Recipe per case: weights = synth_state_dict(reference_model.state_dict(), seed=W_SEED),
input = synth_clips(..., seed=X_SEED); outputs from `model.features` / `model(x)` in eval mode.
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import ref_shim, tv_standin  # noqa: E402
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import synth_clips, synth_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
W_SEED, X_SEED = 1234, 99

# name -> (input shape, factory kwargs)
CASES = {
    "resnet3d50_cfg2": ("resnet3d50", (8, 3, 16, 224, 224), dict(num_classes=339, pretrained=None)),
    "resnet3d50_small": ("resnet3d50", (2, 3, 8, 64, 64), dict(num_classes=339, pretrained=None)),
    "resnet3d50_odd": ("resnet3d50", (3, 3, 5, 50, 70), dict(num_classes=17, pretrained=None)),
    "resnet3d10_small": ("resnet3d10", (2, 3, 8, 64, 64), dict()),
    "resnet3d18_small": ("resnet3d18", (2, 3, 8, 64, 64), dict(num_classes=400, pretrained=None)),
    "resnet3d34_small": ("resnet3d34", (1, 3, 4, 48, 48), dict(num_classes=400, pretrained=None)),
    "nonlocalresnet3d50_small": ("nonlocalresnet3d50", (2, 3, 8, 64, 64), dict(pretrained=None)),
    "r2plus1d18_small": ("r2plus1d18", (2, 3, 8, 64, 64), dict(num_classes=174)),
    "r2plus1d50_small": ("r2plus1d50", (2, 3, 8, 64, 64), dict(num_classes=400)),
    "nonlocal_r2plus1d50_small": ("nonlocal_r2plus1d50", (2, 3, 8, 64, 64), dict(num_classes=339)),
    "resnet18_cfg1": ("resnet18", (1, 3, 224, 224), dict(num_classes=1000, pretrained=None)),
    "resnet50_2d_small": ("resnet50", (3, 3, 96, 64), dict(num_classes=339, pretrained=None)),
    "resnext3d50_small": ("resnext3d50", (2, 3, 8, 64, 64), dict(num_classes=400)),
    "resnext3d10_odd": ("resnext3d10", (3, 3, 5, 50, 70), dict(num_classes=17)),
    "resnext3d50_full": ("resnext3d50", (2, 3, 16, 224, 224), dict(num_classes=400)),
    "wideresnet3d50_small": ("wideresnet3d50", (2, 3, 8, 64, 64), dict(num_classes=400, pretrained=None)),
    "preact_resnet3d50_small": ("preact_resnet3d50", (2, 3, 8, 64, 64), dict(num_classes=339)),
    "preact_resnet3d18_odd": ("preact_resnet3d18", (3, 3, 5, 50, 70), dict(num_classes=17, shortcut_type="A")),
    # multi-view ResNets (multiview.py: module-level upstream, imported through the F6 shim)
    "mvresnet18_small": ("mvresnet18", (2, 3, 8, 64, 64), dict(num_classes=339)),
    "mvresnet50_small": ("mvresnet50", (2, 3, 8, 64, 64), dict(num_classes=174)),
    "mvresnet10_odd": ("mvresnet10", (3, 3, 5, 50, 70), dict(num_classes=17)),
    # BASELINE.json config 3 at full size (8 x 3 x 32 x 112 x 112): the composite and its two parents
    "nonlocal_r2plus1d50_cfg3": ("nonlocal_r2plus1d50", (8, 3, 32, 112, 112), dict(num_classes=339)),
    "r2plus1d50_cfg3": ("r2plus1d50", (8, 3, 32, 112, 112), dict(num_classes=400)),
    "nonlocalresnet3d50_cfg3": ("nonlocalresnet3d50", (8, 3, 32, 112, 112), dict(pretrained=None)),
    # round 4 (VERDICT r3 #1c): the NL branch at FULL strength (W.1 gamma undamped) at the reference's sequence lengths --
    # N = 1568 / 196 keys for config 3, N = 3136 / 392 at the reference's usual 16 x 224 x 224 input (nonlocalnet.py:594-619)
    "nonlocal_r2plus1d50_cfg3_fullnl": ("nonlocal_r2plus1d50", (8, 3, 32, 112, 112), dict(num_classes=339)),
    "nonlocalresnet3d50_16x224_fullnl": ("nonlocalresnet3d50", (2, 3, 16, 224, 224), dict(pretrained=None)),
}
# per-case BN damping of the synthetic-weights recipe (calibrated so max|logit| lands in 10-30 at
# that input size); cases not listed use the defaults of synth_state_dict
RECIPES = {
    # nl 0.05: with the default 0.2 this network's OWN fp32 noise floor (reference CPU fp32 vs an fp64
    # evaluation) is 1.3e-3 -- above the 1e-3 bar; 0.05 brings it to 1.2e-5
    "nonlocal_r2plus1d50_cfg3": dict(inner_bn_damp=0.9, nl_bn_damp=0.05),
    "r2plus1d50_cfg3": dict(inner_bn_damp=0.9),
    "resnet50_2d_small": dict(last_bn_damp=0.7),
    # grouped 3x3x3 convs under fan_out init shrink the signal: un-damp the closing BNs instead
    "resnext3d50_small": dict(last_bn_damp=2.0),
    "resnext3d10_odd": dict(last_bn_damp=2.0),
    "resnext3d50_full": dict(last_bn_damp=2.0),
    "nonlocalresnet3d50_cfg3": dict(last_bn_damp=0.65, nl_bn_damp=0.05),
    # full-strength NL: what made nl 0.2 ill-conditioned is not the branch's strength but its softmax INPUT -- theta^T phi
    # reaches 1e2 ... 1e4 with kaiming embeddings (a hard arg-max over 1568 keys whose winner flips under fp32 rounding).
    # Damping theta / phi per stage brings the affinities to 0.7 ... 60 (soft to moderately peaked attention, as in a
    # trained network); then the branch can run undamped and the fp32 noise floor stays at 7e-6 / 3e-5.
    "nonlocal_r2plus1d50_cfg3_fullnl": dict(inner_bn_damp=0.85, last_bn_damp=0.55, nl_bn_damp=1.0,
                                            nl_embed_damp={"layer2": 0.3, "layer3": 0.1}),
    "nonlocalresnet3d50_16x224_fullnl": dict(last_bn_damp=0.4, nl_bn_damp=1.0, nl_embed_damp={"layer2": 0.2, "layer3": 0.08}),
    # MultiViewConv: a 1x1x1 conv's three views are identical, so it gains sum(a) ~ 1.8 per layer: damp per block
    "mvresnet50_small": dict(last_bn_damp=0.19),
    "mvresnet18_small": dict(last_bn_damp=0.55),
    "mvresnet10_odd": dict(last_bn_damp=1.1),
}


def build_composite(ref, r2):
    """(2+1)D bottlenecks + NL blocks inside NonLocalResNet3D, shortcut 'B'."""
    nl = ref.models.nonlocalnet

    class NLBottleneck2p1d(nl.NonLocalBottleneck):
        Conv3d = r2.SpatioTemporalConv

    class NLR2Plus1D(nl.NonLocalResNet3D):
        Conv3d = r2.SpatioTemporalConv

        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.conv1 = r2.SpatioTemporalConv(3, 64, 7, stride=(1, 2, 2), padding=(3, 3, 3), bias=False)

        def init_weights(self):
            r2.R2Plus1D.init_weights(self)

    return lambda num_classes=339: NLR2Plus1D(NLBottleneck2p1d, [3, 4, 6, 3], [0, 2, 3, 0],
                                              shortcut_type="B", num_classes=num_classes)


# TRN wrapper fixtures: name -> (TRN kwargs, input [B,T,3,H,W], numpy seed for MSTRN's subset draw)
TRN_CASES = {
    "trn_htrn_small": (dict(num_classes=51, num_segments=4, consensus="HTRN"), (2, 4, 3, 64, 64), None),
    "trn_mstrn_small": (dict(num_classes=51, num_segments=4, consensus="MSTRN", frame_bottleneck_dim=256,
                             video_feature_dim=128), (2, 4, 3, 64, 64), 11),
    "trn_trn_b1": (dict(num_classes=17, num_segments=3, consensus="TRN", frame_bottleneck_dim=128,
                        video_feature_dim=64), (1, 3, 3, 64, 96), None),
}


# SlowFast (slowfast.py): name -> (factory, mode, kwargs, input shape)
SLOWFAST_CASES = {
    "slowfast50_sf_small": ("resnet50", "SF", dict(num_classes=51), (2, 3, 32, 64, 64)),
    "slowfast50_s_small": ("resnet50", "S", dict(num_classes=51), (2, 3, 32, 64, 64)),
    "slowfast50_f_small": ("resnet50", "F", dict(num_classes=51), (2, 3, 32, 64, 64)),
    "slowfast18_sf_small": ("resnet18", "SF", dict(num_classes=51), (3, 3, 32, 48, 80)),
    "slowfast50_sf_full": ("resnet50", "SF", dict(num_classes=400), (2, 3, 64, 224, 224)),
}


def make_slowfast(ref, keys_out, only):
    for case, (fac, mode, kw, shape) in SLOWFAST_CASES.items():
        if only and case not in only and "slowfast" not in only:
            continue
        model = getattr(ref.slowfast, fac)(mode=mode, **kw)
        model.eval()
        sd = synth_state_dict(model.state_dict(), W_SEED)
        model.load_state_dict(sd)
        keys_out[case] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(X_SEED))
        with torch.no_grad():
            logits = model(x)
        np.savez_compressed(os.path.join(OUT, case + ".npz"), logits=logits.numpy(), shape=np.array(shape),
                            w_seed=W_SEED, x_seed=X_SEED, factory=np.array(fac), mode=np.array(mode),
                            kwargs=np.array(json.dumps(kw)))
        print("%-28s logits %s max|.|=%.3f argmax=%s" % (case, tuple(logits.shape), logits.abs().max().item(),
                                                        logits.argmax(1).tolist()))


def build_ref_trn(ref, trn, **kw):
    """The reference TRN (trn.py:194-244) cannot be constructed without a download: it reads
    base_model.mean/std (set only by load_pretrained).  Here `pretrainedmodels.__dict__[arch]`
    resolves to the reference's own 2-D wrapper (torchvision_models.py:506-514, over the torchvision
    stand-in) built with pretrained=None and given the 'moments' settings by hand."""
    tvm = ref.models.torchvision_models

    def make(arch):
        def factory(num_pc, pretrained):
            m = ref.__dict__[arch](num_classes=num_pc, pretrained=None)
            st = tvm.pretrained_settings[arch].get("moments", tvm.pretrained_settings[arch]["imagenet"])
            m.input_space, m.input_size, m.input_range = st["input_space"], st["input_size"], st["input_range"]
            m.mean, m.std = st["mean"], st["std"]
            return m
        return factory

    trn.pretrainedmodels = types.SimpleNamespace(**{a: make(a) for a in tv_standin.FACTORIES})
    return trn.TRN(**kw)


def main():
    only = set(sys.argv[1:])
    torch.manual_seed(0)
    ref = ref_shim.import_reference(tv_standin.FACTORIES)
    r2 = ref_shim.import_r2plus1d()
    trn = ref_shim.import_trn()
    composite = build_composite(ref, r2)
    keys_path = os.path.join(OUT, "state_keys.json")
    keys_out = json.load(open(keys_path)) if (only and os.path.exists(keys_path)) else {}

    # (2+1)D reference models must be *built and run* before any resnet3d* factory patches
    # ResNet3D.forward at class level -- so handle them first.
    order = sorted(CASES, key=lambda n: 0 if ("r2plus1d" in n or "preact" in n or "mvresnet" in n) else 1)
    for case in order:
        if only and case not in only:
            continue
        arch, shape, kw = CASES[case]
        if arch == "nonlocal_r2plus1d50":
            model = composite(**kw)
        elif arch.startswith("r2plus1d"):
            model = getattr(r2, arch)(**kw)
        elif arch.startswith("preact"):
            model = getattr(ref_shim.import_preact(), arch)(**kw)
        elif arch.startswith("wideresnet"):
            model = getattr(ref_shim.import_wideresnet3d(), arch)(**kw)
        elif arch.startswith("mvresnet"):
            model = getattr(ref_shim.import_multiview(), arch)(**kw)
        else:
            model = ref.__dict__[arch](**kw)
        model.eval()
        recipe = RECIPES.get(case, {})
        sd = synth_state_dict(model.state_dict(), W_SEED, **recipe)
        model.load_state_dict(sd)
        keys_out[case] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        x = synth_clips(shape[0], shape[2], shape[3], X_SEED) if len(shape) == 5 else None
        if len(shape) == 5 and shape[3] != shape[4]:
            g = torch.Generator().manual_seed(X_SEED)
            x = torch.randn(*shape, generator=g)
        if len(shape) == 4:
            g = torch.Generator().manual_seed(X_SEED)
            x = torch.randn(*shape, generator=g)
        with torch.no_grad():
            if hasattr(model, "features") and not arch.startswith(("r2plus1d", "resnext", "wideresnet", "preact", "mvresnet")):
                feat = model.features(x)
                logits = model.logits(feat)
            else:   # R2Plus1D keeps ResNet3D.forward / fc (r2plus1d.py:99-110)
                feat = None
                logits = _r2_forward(model, x)
        blob = dict(logits=logits.numpy(), shape=np.array(shape), w_seed=W_SEED, x_seed=X_SEED,
                    recipe=np.array(json.dumps(recipe)))
        if case.endswith("_cfg3") or case.endswith("_cfg2") or case.endswith("_fullnl"):
            # conditioning of the fixture: fp32 reference vs an fp64 evaluation of the oracle (2 clips)
            from oracle import functional as OF
            arch_cfg = OF.ARCHS[arch]
            sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
            y64 = OF.forward(arch_cfg, sd64, x[:2].double())
            blob["fp32_noise_floor"] = np.float64((logits[:2].double() - y64).abs().max().item())
            print("   fp32 noise floor (reference fp32 vs fp64, 2 clips): %.2e" % blob["fp32_noise_floor"])
        if feat is not None:
            f = feat.numpy()
            blob["feat_shape"] = np.array(f.shape)
            blob["feat_sum"] = np.float64(f.astype(np.float64).sum())
            blob["feat_abs_sum"] = np.float64(np.abs(f.astype(np.float64)).sum())
            if f.size <= 300000:
                blob["features"] = f
        np.savez_compressed(os.path.join(OUT, case + ".npz"), **blob)
        print("%-28s logits %s max|.|=%.3f argmax=%s" % (case, tuple(logits.shape), logits.abs().max().item(),
                                                        logits.argmax(1).tolist()))

    if not only or "trn" in only:
        make_trn(ref, trn, keys_out)
    if not only or "nlblock" in only:
        make_nlblock(ref, keys_out)
    if not only or "mnist_nl" in only:
        make_mnist_nl(ref, keys_out)
    if not only or any(c.startswith("slowfast") for c in only):
        make_slowfast(ref, keys_out, only)
    json.dump(keys_out, open(keys_path, "w"))


NL_CASES = [("embedded_gaussian", False, True), ("embedded_gaussian", True, True), ("dot_product", False, True),
            ("dot_product", True, False), ("gaussian", False, True), ("gaussian", True, False),
            ("concatenation", False, True), ("concatenation", True, False)]


def make_mnist_nl(ref, keys_out):
    """MNISTNonLocalNet (nonlocalnet.py:273-309) on a batch of 28x28 single-channel images."""
    net = ref.models.nonlocalnet.MNISTNonLocalNet().eval()
    sd = synth_state_dict(net.state_dict(), W_SEED, nl_bn_damp=1.0)
    net.load_state_dict(sd)
    x = torch.randn(6, 1, 28, 28, generator=torch.Generator().manual_seed(X_SEED))
    keys_out["mnist_nl"] = [[k, list(v.shape)] for k, v in net.state_dict().items()]
    with torch.no_grad():
        y = net(x)
    np.savez_compressed(os.path.join(OUT, "mnist_nl.npz"), logits=y.numpy(), shape=np.array(x.shape), w_seed=W_SEED,
                        x_seed=X_SEED, recipe=np.array(json.dumps(dict(nl_bn_damp=1.0))))
    print("mnist_nl logits", tuple(y.shape), "max|.|=%.3f" % y.abs().max().item(), y.argmax(1).tolist())


def make_nlblock(ref, keys_out):
    """Standalone NonLocalBlock3D (nonlocalnet.py:264-270) in every mode the HIP path runs."""
    nl = ref.models.nonlocalnet
    x = torch.randn(2, 16, 4, 8, 6, generator=torch.Generator().manual_seed(X_SEED))
    blob = dict(shape=np.array(x.shape), w_seed=W_SEED, x_seed=X_SEED)
    for mode, sub, bn in NL_CASES:
        blk = nl.NonLocalBlock3D(16, mode=mode, sub_sample=sub, bn_layer=bn).eval()
        sd = synth_state_dict(blk.state_dict(), W_SEED)
        blk.load_state_dict(sd)
        tag = "%s_%d_%d" % (mode, sub, bn)
        keys_out["nlblock_" + tag] = [[k, list(v.shape)] for k, v in blk.state_dict().items()]
        with torch.no_grad():
            blob[tag] = blk(x).numpy()
    np.savez_compressed(os.path.join(OUT, "nlblock.npz"), **blob)
    print("nlblock modes", [k for k in blob if k not in ("shape", "w_seed", "x_seed")])
    # NonLocalBlock2D / NonLocalBlock1D (nonlocalnet.py:246-261): the same block over [B,C,H,W] / [B,C,L]
    for dim, cls, shape in ((2, "NonLocalBlock2D", (2, 16, 10, 12)), (1, "NonLocalBlock1D", (3, 16, 30))):
        x = torch.randn(*shape, generator=torch.Generator().manual_seed(X_SEED))
        blob = dict(shape=np.array(x.shape), w_seed=W_SEED, x_seed=X_SEED)
        for mode, sub, bn in NL_CASES:
            blk = getattr(nl, cls)(16, mode=mode, sub_sample=sub, bn_layer=bn).eval()
            sd = synth_state_dict(blk.state_dict(), W_SEED)
            blk.load_state_dict(sd)
            tag = "%s_%d_%d" % (mode, sub, bn)
            keys_out["nlblock%dd_%s" % (dim, tag)] = [[k, list(v.shape)] for k, v in blk.state_dict().items()]
            with torch.no_grad():
                blob[tag] = blk(x).numpy()
        np.savez_compressed(os.path.join(OUT, "nlblock%dd.npz" % dim), **blob)
        print("nlblock%dd modes" % dim, len(NL_CASES))


def make_trn(ref, trn, keys_out):
    # TRN relation heads (standalone nn.Modules)
    g = torch.Generator().manual_seed(X_SEED)
    xr = torch.randn(4, 1, 8, 256, generator=g)
    rel = trn.Relation(8, 256, 96, 128).eval()
    sd = synth_state_dict(rel.state_dict(), W_SEED)
    rel.load_state_dict(sd)
    keys_out["relation"] = [[k, list(v.shape)] for k, v in rel.state_dict().items()]
    with torch.no_grad():
        yr = rel(xr)
    msr = trn.MultiScaleRelation(8, 256, 96, 128, 3).eval()
    sd = synth_state_dict(msr.state_dict(), W_SEED)
    msr.load_state_dict(sd)
    keys_out["multiscale_relation"] = [[k, list(v.shape)] for k, v in msr.state_dict().items()]
    np.random.seed(7)
    with torch.no_grad():
        ym = msr(xr)
    np.savez_compressed(os.path.join(OUT, "trn_relation.npz"), relation=yr.numpy(), multiscale=ym.numpy(),
                        np_seed=7, w_seed=W_SEED, x_seed=X_SEED)
    print("trn relation", tuple(yr.shape), "multiscale", tuple(ym.shape))

    # HierarchicalRelation at the only depth the reference forward runs (0), as TRN builds it
    hr = trn.HierarchicalRelation(8, 256, 96, 1024).eval()
    sd = synth_state_dict(hr.state_dict(), W_SEED)
    hr.load_state_dict(sd)
    keys_out["hierarchical_relation"] = [[k, list(v.shape)] for k, v in hr.state_dict().items()]
    with torch.no_grad():
        yh = hr(xr)
    np.savez_compressed(os.path.join(OUT, "trn_hierarchical.npz"), out=yh.numpy(), w_seed=W_SEED, x_seed=X_SEED)

    # the TRN wrapper (trn.py:194-263) over the reference's 2-D resnet50 wrapper (torchvision stand-in)
    for case, (kw, shape, np_seed) in TRN_CASES.items():
        model = build_ref_trn(ref, trn, **kw)
        model.eval()                       # (the reference's TRN.train override returns None)
        sd = synth_state_dict(model.state_dict(), W_SEED)
        model.load_state_dict(sd)
        keys_out[case] = [[k, list(v.shape)] for k, v in model.state_dict().items()]
        g = torch.Generator().manual_seed(X_SEED)
        x = torch.randn(*shape, generator=g)
        if np_seed is not None:
            np.random.seed(np_seed)
        with torch.no_grad():
            feat = model.features(x)
            logits = model.logits(feat)
        np.savez_compressed(os.path.join(OUT, case + ".npz"), logits=logits.numpy(), features=feat.numpy(),
                            shape=np.array(shape), w_seed=W_SEED, x_seed=X_SEED,
                            np_seed=-1 if np_seed is None else np_seed, kwargs=np.array(json.dumps(kw)))
        print("%-28s logits %s max|.|=%.3f features %s" % (case, tuple(logits.shape), logits.abs().max().item(),
                                                          tuple(feat.shape)))


def _r2_forward(model, x):
    """ResNet3D.forward as written at resnet3D.py:203-218 (the class attribute may have been
    replaced by modify_resnets in this process, F7): stem, pool, stages, avgpool, fc."""
    h = model.maxpool(model.relu(model.bn1(model.conv1(x))))
    h = model.layer4(model.layer3(model.layer2(model.layer1(h))))
    h = model.avgpool(h)
    head = model.fc if getattr(model, "fc", None) is not None else model.last_linear
    return head(h.view(h.size(0), -1))


if __name__ == "__main__":
    main()
