"""The workloads bench.py can run: BASELINE.json's configurations (config 2 is the headline and the
default) plus a CPU stand-in used only by the multi-rank functional check (PTX_BENCH_STANDIN=1).

A workload is the tuple
    (model, weight recipe, make(n, seed) -> n CPU units, units per GPU, forward(model, device input) or None,
     cpu oracle fn(sd, units, idx), unit, label, parity sample indices or None)
`idx`: positions of the sampled units inside this rank's batch (labels of config 5 follow them).
"""
import torch

CLIPS_PER_GPU, FRAMES, SIZE, CLASSES = 8, 16, 224, 339
GFLOP_PER_CLIP = 79.692           # 2 x 318.768 GMAC / 8 clips, padding taps counted
PEAK_F32_MFMA_TF = 157.3          # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_F16_MFMA_TF = 2500.0         # MI355X_MICROARCH.md "Peak BF16/FP16 MFMA" (dense)
SUSTAINED_F16_MFMA_TF = 1600.0    # measured: scripts/micro/mfma_f16_peak.hip, random operands (profiles/r03_mfma_f16_peak.txt)
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md HBM3E peak

# BASELINE.json configs: the batch the metric is quoted on (what --scaling strong shards)
GLOBAL_BATCH = {"cfg1": 1, "cfg2": 8, "cfg3": 8, "cfg4": 16}


def _randn(*shape):
    return lambda n, seed: torch.randn(n, *shape, generator=torch.Generator().manual_seed(seed))


def headline_workload():
    """Config 2: resnet3d50 (Moments-339), 8 x 3 x 16 x 224 x 224 synthetic clips per GPU."""
    import pretorched_x_amd as ptx
    from oracle import functional as OF
    from pretorched_x_amd.testing import synth_clips
    model = ptx.__dict__["resnet3d50"](num_classes=CLASSES, pretrained=None)
    label = ("resnet3d50 (Moments-339) forward, %dx3x%dx%dx%d synthetic clips per GPU, "
             "random-init weights (seeded recipe)" % (CLIPS_PER_GPU, FRAMES, SIZE, SIZE))
    return (model, {}, (lambda n, seed: synth_clips(n, FRAMES, SIZE, seed)), CLIPS_PER_GPU, None,
            (lambda sd, x, idx: OF.forward(OF.ARCHS["resnet3d50"], sd, x)), "clips", label, None)


def other_workload(name, rank):
    """The non-headline BASELINE.json configurations (same tuple as headline_workload)."""
    import pretorched_x_amd as ptx
    from oracle import functional as OF
    from pretorched_x_amd.testing import BIGGAN_RECIPE, I3D_RECIPE

    if name == "cfg1":
        m = ptx.resnet18(num_classes=1000, pretrained=None)
        return m, {}, _randn(3, 224, 224), 1, None, lambda sd, x, idx: OF.forward(OF.ARCHS["resnet18"], sd, x), "images", \
            "resnet18 2-D forward, 1x3x224x224 (config 1; arithmetic reference = torchvision stand-in, parity unpinned)", None
    if name == "cfg3":
        m, recipe = ptx.nonlocal_r2plus1d50(339), dict(inner_bn_damp=0.9, nl_bn_damp=0.05)
        return m, recipe, _randn(3, 32, 112, 112), 8, None, \
            lambda sd, x, idx: OF.forward(OF.ARCHS["nonlocal_r2plus1d50"], sd, x), "clips", \
            "resnet2p1d50 + NL blocks forward, 8x3x32x112x112 synthetic clips per GPU (config 3)", None
    if name == "cfg4":
        from oracle import i3d_standin as I3
        m, recipe = ptx.i3d(400), I3D_RECIPE
        return m, recipe, _randn(3, 64, 224, 224), 2, None, lambda sd, x, idx: I3.forward(sd, x), "clips", \
            "I3D (InceptionV1-3D) forward, 2x3x64x224x224 synthetic clips per GPU = 16 over 8 GPUs (config 4; parity unpinned)", None
    if name in ("cfg5", "cfg5-fp32"):
        from oracle import biggan_standin as BG
        half = name == "cfg5"
        m, recipe = ptx.biggan_deep(256, precision="fp16" if half else "fp32"), BIGGAN_RECIPE
        g = torch.Generator().manual_seed(99 + rank)
        z = torch.randn(64, 128, generator=g)
        lab = torch.randint(0, 1000, (64,), generator=g)

        def fwd(model, zd, lab=lab):
            return model(zd, model.shared(lab.to(zd.device)))
        # Engine.generate runs batch 64 as two 32-image chunks: the parity sample takes images from BOTH
        return m, recipe, (lambda n, seed: z[:n]), 64, fwd, \
            lambda sd, zs, idx: BG.forward(sd, zs, sd["shared.weight"][lab[idx]]), "images", \
            ("BigGAN-deep-256 generator, batch 64 z ~ N(0,1) + class labels per GPU, %s (config 5; parity unpinned)" %
             ("fp16 MFMA operands, fp32 accumulate / skip / output" if half else "fp32 MFMA path")), [0, 31, 32, 63]
    raise SystemExit("unknown workload %r" % name)


class StandInNet(torch.nn.Module):
    """A few-kFLOP CPU network with the path's SHAPE CONTRACT only (clips [B,3,T,H,W] -> logits [B, classes]): what the
    ranks of the multi-rank functional check (PTX_BENCH_STANDIN=1, gloo, no GPU) run instead of the HIP engine.  Never a
    measurement: the line such a run prints says so in `config.parallelism` and carries no roofline."""

    def __init__(self, classes=CLASSES):
        super().__init__()
        g = torch.Generator().manual_seed(4321)
        self.w = torch.nn.Parameter(torch.randn(classes, 3, generator=g), requires_grad=False)

    def forward(self, x):
        return x.float().mean(dim=(2, 3, 4)) @ self.w.t()


def standin_workload(name):
    """The stand-in for workload `name`: tiny clips (3 x 2 x 4 x 4), the REAL per-GPU / global batch sizes of that
    configuration, so the sharding, gather and per-rank bookkeeping see the shapes a real 8-GPU run has."""
    per_gpu = {"cfg1": 1, "cfg2": CLIPS_PER_GPU, "cfg3": 8, "cfg4": 2}.get(name)
    if per_gpu is None:
        raise SystemExit("PTX_BENCH_STANDIN: no clip batch to shard for %s" % name)
    m = StandInNet()
    return (m, None, _randn(3, 2, 4, 4), per_gpu, None, (lambda sd, x, idx: m(x)), "clips",
            "STAND-IN forward (CPU, a few kFLOP) with %s's batch sizes -- functional check of the multi-rank plumbing" % name, None)
