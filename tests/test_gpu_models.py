"""GPU parity tests, model level: the HIP engine behind the reference's model API against
(a) golden outputs of the real reference (tests/golden/*.npz, generated in the builder container)
and (b) the CPU oracle restatement run in the same process on the same seeded weights/inputs.

Bar (BASELINE.json north_star): |logits_gpu - logits_cpu| <= 1e-3 in fp32 and identical argmax -- applied
unscaled (round 1 widened it by max|logit| / 30; measured errors are 1e-5 class, so the plain bar holds).
"""
import os

import numpy as np
import pytest
import torch

from conftest import (GOLDEN_CASES, SLOWFAST_CASES, TRN_CASES, golden_input, golden_recipe, golden_slowfast,
                      golden_trn, load_golden, oracle_cfg)
from oracle import functional as OF
from pretorched_x_amd.testing import synth_clips, synth_state_dict

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-3


def _build(ptx, arch, kw, seed, **recipe):
    model = ptx.__dict__[arch](**kw)
    sd = synth_state_dict(model.state_dict(), seed, **recipe)
    model.load_state_dict(sd)
    return model.to(DEV).eval(), sd


def _ftol(want):
    """Bound for FEATURE maps (not logits; magnitudes reach 100+): 3e-5 of the map's scale (the (2+1)D + NL composite's
    own fp32 reorder noise is 1e-5 of it), never below the logits bar."""
    return max(TOL, 3e-5 * want.abs().max().item())


def _check(got, want, what, tol=TOL):
    """The stated bar, unscaled: max |got - want| <= 1e-3 (BASELINE.json north_star) -- for logits."""
    got = got.detach().cpu()
    err = (got - want).abs().max().item()
    assert got.shape == want.shape, what
    assert err <= tol, "%s: max abs err %.3e > %.1e (max|ref| %.2f)" % (what, err, tol, want.abs().max().item())
    return err


from conftest import FULL_SIZE
SMALL = [c for c in GOLDEN_CASES if c not in FULL_SIZE]


@pytest.mark.parametrize("case", SMALL)
def test_model_parity_small(ptx, case):
    arch, kw = GOLDEN_CASES[case]
    blob = load_golden(case)
    model, sd = _build(ptx, arch, kw, **golden_recipe(blob))
    x = golden_input(blob)
    xd = x.to(DEV)
    feats = model.features(xd)
    logits = model.logits(feats)
    fwd = model(xd)
    torch.cuda.synchronize()
    # API contract: forward == logits(features)
    assert (fwd - logits).abs().max().item() <= 1e-5 * max(1.0, logits.abs().max().item())
    # (a) golden = the real reference's output
    ref_logits = torch.from_numpy(blob["logits"])
    _check(logits, ref_logits, case + " logits vs golden")
    _check(fwd, ref_logits, case + " forward vs golden")
    assert torch.equal(fwd.cpu().argmax(1), ref_logits.argmax(1))
    if "features" in blob.files:
        gf = torch.from_numpy(blob["features"])
        _check(feats, gf, case + " features vs golden", _ftol(gf))
    # (b) oracle restatement, same run
    cfg = oracle_cfg(arch, kw)
    with torch.no_grad():
        of = OF.features(cfg, sd, x)
        ol = OF.logits(cfg, sd, of)
    _check(feats, of, case + " features vs oracle", _ftol(of))
    _check(logits, ol, case + " logits vs oracle")
    assert feats.is_contiguous() and tuple(feats.shape) == tuple(of.shape)


def test_config2_full_size_parity(ptx):
    """The headline configuration: resnet3d50 (339 classes), 8x3x16x224x224 clips."""
    blob = load_golden("resnet3d50_cfg2")
    model, sd = _build(ptx, "resnet3d50", dict(num_classes=339, pretrained=None), int(blob["w_seed"]))
    x = synth_clips(8, 16, 224, int(blob["x_seed"]))
    out = model(x.to(DEV))
    torch.cuda.synchronize()
    ref = torch.from_numpy(blob["logits"])
    err = _check(out, ref, "cfg2 logits vs golden")
    assert torch.equal(out.cpu().argmax(1), ref.argmax(1))
    feats = model.features(x.to(DEV))
    f = feats.double()
    assert tuple(feats.shape) == tuple(int(v) for v in blob["feat_shape"])
    assert abs(f.sum().item() - float(blob["feat_sum"])) <= 1e-4 * float(blob["feat_abs_sum"])
    assert abs(f.abs().sum().item() - float(blob["feat_abs_sum"])) <= 1e-4 * float(blob["feat_abs_sum"])
    print("cfg2 max|dlogits| = %.3e (max|logit| %.2f)" % (err, ref.abs().max().item()))


@pytest.mark.parametrize("case", [c for c in FULL_SIZE if c != "resnet3d50_cfg2"])
def test_config3_full_size_parity(ptx, case):
    """BASELINE.json config 3 (8x3x32x112x112): (2+1)D + non-local composite and its two parents,
    against the real reference's logits (golden) at full size."""
    arch, kw = GOLDEN_CASES[case]
    blob = load_golden(case)
    model, _ = _build(ptx, arch, kw, **golden_recipe(blob))
    x = golden_input(blob)
    out = model(x.to(DEV))
    torch.cuda.synchronize()
    ref = torch.from_numpy(blob["logits"])
    err = _check(out, ref, case + " logits vs golden")
    assert torch.equal(out.cpu().argmax(1), ref.argmax(1))
    print("%s max|dlogits| = %.3e (max|logit| %.2f)" % (case, err, ref.abs().max().item()))


@pytest.mark.parametrize("case", ["resnet3d50_cfg2", "nonlocal_r2plus1d50_cfg3"])
def test_strong_scaling_shard_shapes_against_the_goldens(ptx, case):
    """at 8 GPUs the metric's 8-clip batch leaves ONE clip per rank (config 4: two) -- plans of batch 1 and
    batch 2 pick other tiles / split-K factors than the 8-clip plan and had no parity test.  Clips are independent
    (eval-mode BN, per-sample attention), so the first rows of the full-size goldens are the expected logits of the shards
    rank 0 would run: clip 0 alone, clips 0-1, and -- the shard of the LAST rank -- clip 7 alone."""
    blob = load_golden(case)
    ref = torch.from_numpy(blob["logits"])
    if case == "resnet3d50_cfg2":
        model, _ = _build(ptx, "resnet3d50", dict(num_classes=339, pretrained=None), int(blob["w_seed"]))
        x = synth_clips(8, 16, 224, int(blob["x_seed"]))
    else:
        arch, kw = GOLDEN_CASES[case]
        model, _ = _build(ptx, arch, kw, **golden_recipe(blob))
        x = golden_input(blob)
    assert x.shape[0] == 8 and ref.shape[0] == 8
    for lo, hi in ((0, 1), (0, 2), (7, 8)):
        out = model(x[lo:hi].contiguous().to(DEV))
        torch.cuda.synchronize()
        err = _check(out, ref[lo:hi], "%s clips [%d, %d) vs golden rows" % (case, lo, hi))
        assert torch.equal(out.cpu().argmax(1), ref[lo:hi].argmax(1))
        print("%s shard [%d, %d): max|dlogits| = %.3e" % (case, lo, hi, err))
    assert len(model.engine()._plans) == 2          # one plan per shard SHAPE (batch 1, batch 2)


def test_stream_k_attention_in_a_plan(ptx, monkeypatch):
    """PTX_NL_STREAMK=1 (off by default: measured slower): the N = 1568 attention launches of config 3's
    full-strength NL composite run the stream-K form over the plan's scratch buffer -- same golden logits at the 1e-3 bar,
    1e-4-class equal to the plain kernels, and the plan asks for the scratch only then."""
    case = "nonlocal_r2plus1d50_cfg3_fullnl"
    arch, kw = GOLDEN_CASES[case]
    blob = load_golden(case)
    model, _ = _build(ptx, arch, kw, **golden_recipe(blob))
    x = golden_input(blob).to(DEV)
    ref = torch.from_numpy(blob["logits"])
    plain = model(x).clone()
    assert all(p_.nl_ws_bytes == 0 for p_ in model.engine()._plans.values())
    monkeypatch.setenv("PTX_NL_STREAMK", "1")
    model.engine().invalidate()
    out = model(x)
    plan = list(model.engine()._plans.values())[-1]
    assert plan.nl_ws_bytes == x.shape[0] * 32 * 2 * (64 * 256 + 128) * 4 and plan.nl_ws is not None
    _check(out, ref, case + " (stream-K attention) logits vs golden")
    assert torch.equal(out.cpu().argmax(1), ref.argmax(1))
    assert (out - plain).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    assert torch.equal(model(x), out)
    monkeypatch.delenv("PTX_NL_STREAMK")
    model.engine().invalidate()
    assert torch.equal(model(x), plain)


def test_full_size_properties(ptx):
    """Size-independent properties at the full config-2 shape: determinism (bit-exact),
    batch-permutation equivariance (split-K slices the tile's *pruned* k-space, so the grouping of
    partial sums depends on where a clip's rows fall in a tile: fp32 reorder noise only) and
    agreement of B=8 with B=1 plans (different tiles / split-K)."""
    model, _ = _build(ptx, "resnet3d50", dict(num_classes=339, pretrained=None), 77)
    x = synth_clips(8, 16, 224, 5).to(DEV)
    a = model(x)
    b = model(x)
    assert torch.equal(a, b)
    perm = torch.tensor([3, 0, 7, 1, 6, 2, 5, 4], device=DEV)
    c = model(x[perm].contiguous())
    assert (c - a[perm]).abs().max().item() <= 2e-5 * max(1.0, a.abs().max().item())
    assert torch.equal(c.argmax(1), a[perm].argmax(1))
    one = model(x[2:3].contiguous())
    assert (one - a[2:3]).abs().max().item() <= 1e-4 * max(1.0, a.abs().max().item())


def test_last_linear_contract_and_weight_updates(ptx):
    model, sd = _build(ptx, "resnet3d50", dict(num_classes=339, pretrained=None), 3)
    x = synth_clips(2, 8, 64, 9)
    xd = x.to(DEV)
    base = model(xd).cpu()
    cfg = OF.ARCHS["resnet3d50"]
    with torch.no_grad():
        feat = OF.features(cfg, sd, x)
    pooled = torch.nn.functional.adaptive_avg_pool3d(feat, 1).flatten(1)
    # users replace last_linear by an Identity (README "last_linear") ...
    keep = model.last_linear
    model.last_linear = torch.nn.Identity()
    _check(model(xd), pooled, "identity head (forward)")
    _check(model.logits(model.features(xd)), pooled, "identity head (logits)")
    # ... or by a new Linear with another class count
    torch.manual_seed(0)
    new = torch.nn.Linear(2048, 10).to(DEV)
    model.last_linear = new
    want = torch.nn.functional.linear(pooled, new.weight.detach().cpu(), new.bias.detach().cpu())
    _check(model(xd), want, "new linear head")
    model.last_linear = keep
    assert torch.equal(model(xd).cpu(), base)
    # in-place parameter updates must be picked up (packed weights are refreshed)
    with torch.no_grad():
        model.layer4[2].bn3.weight.mul_(0.5)
    sd2 = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    with torch.no_grad():
        want2 = OF.forward(cfg, sd2, x)
    got2 = model(xd)
    assert (got2.cpu() - base).abs().max().item() > 1e-3
    _check(got2, want2, "after in-place BN update")


def test_input_validation_on_device(ptx):
    model, _ = _build(ptx, "resnet3d10", dict(), 1)
    with pytest.raises(ptx.PtxError):
        model(torch.zeros(1, 3, 4, 32, 32, device=DEV, dtype=torch.float16))
    with pytest.raises(ptx.PtxError):
        model(torch.zeros(1, 3, 32, 32, device=DEV))
    # non-contiguous input is accepted (made contiguous), any T/H/W works (adaptive pool)
    x = synth_clips(1, 6, 40, 2).to(DEV)
    y1 = model(x)
    y2 = model(x.transpose(3, 4).contiguous().transpose(3, 4))
    assert torch.equal(y1, y2)
    # a contiguous view at a storage offset that is not 16-byte aligned (the direct stem DMAs 16-byte pieces of the
    # caller's tensor): one aligned copy, same result; a frame-strided view is just strides
    flat = torch.zeros(x.numel() + 1, device=DEV)
    flat[1:] = x.reshape(-1)
    xv = flat[1:].view_as(x)
    assert xv.data_ptr() % 16 != 0 and xv.is_contiguous()
    assert torch.equal(model(xv), y1) and torch.equal(model.features(xv), model.features(x))
    wide = synth_clips(1, 12, 40, 3).to(DEV)
    assert torch.equal(model(wide[:, :, ::2]), model(wide[:, :, ::2].contiguous()))


def test_oversized_batches_are_split(ptx):
    """Batches whose activations would exceed the 2 GiB per-launch limit are run in slices."""
    model, _ = _build(ptx, "resnet3d10", dict(), 1)
    x = synth_clips(5, 4, 32, 4).to(DEV)
    whole = model(x)
    wf = model.features(x)
    eng = model.engine()
    per_clip = max(a.t.numel() * 4 for a in eng.dry_plan(model, (1, 3, 4, 32, 32)).acts)
    eng.LIMIT_BYTES = 2 * per_clip + 1            # at most 2 clips per launch -> 3 slices for B = 5
    eng.invalidate()
    assert eng.max_batch(model, (3, 4, 32, 32)) == 2
    sliced = model(x)
    sf = model.features(x)
    assert sliced.shape == whole.shape and sf.shape == wf.shape
    assert (sliced - whole).abs().max().item() <= 1e-5 * max(1.0, whole.abs().max().item())
    assert (sf - wf).abs().max().item() <= 1e-5 * max(1.0, wf.abs().max().item())


def test_hipgraph_replay_matches_eager(ptx):
    """Opt-in hipGraph mode: forward() captured once per shape and replayed; same kernels, so the
    logits are bit-identical to the eager launches, also after an in-place weight update (the packed
    filters are refreshed outside the graph, in the buffers the graph reads)."""
    import time
    model, _ = _build(ptx, "resnet3d50", dict(num_classes=339, pretrained=None), 7)
    xs = [synth_clips(1, 8, 64, s).to(DEV) for s in (1, 2, 3)]
    eager = [model(x).clone() for x in xs]
    eng = model.engine()
    eng.use_graph = True
    try:
        for x, want in zip(xs, eager):
            assert torch.equal(model(x), want)
        with torch.no_grad():
            model.layer3[1].bn2.weight.mul_(1.25)
        got = model(xs[0])
        eng.use_graph = False
        assert torch.equal(model(xs[0]), got) and not torch.equal(got, eager[0])
        # launch-bound shape: replay should not be slower than ~90 eager launches
        def bench(n=30):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                model(xs[0])
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        t_eager = bench()
        eng.use_graph = True
        model(xs[0])
        t_graph = bench()
        print("1x3x8x64x64 forward: eager %.3f ms, hipGraph replay %.3f ms" % (t_eager, t_graph))
    finally:
        eng.use_graph = False


def test_clip_lanes_are_slice_forwards_on_their_own_streams(ptx):
    """Engine.lanes = n (opt-in): forward() cuts the batch into n contiguous slices, one plan and one HIP stream each, and
    concatenates the logits in clip order.  A lane IS an ordinary forward of its slice: bit-identical to calling the model on
    that slice (same shape -> same tiles), 1e-3-bar identical to the single-plan forward and to the CPU oracle; batches the
    lane count does not divide, and the hipGraph mode, keep the single-plan path; the caller's stream may be any stream."""
    model, sd = _build(ptx, "resnet3d50", dict(num_classes=17, pretrained=None), 11)
    x_cpu = synth_clips(4, 8, 64, 21)
    x = x_cpu.to(DEV)
    want = OF.forward(OF.ARCHS["resnet3d50"], sd, x_cpu)
    eng = model.engine()
    single = model(x).clone()
    halves = torch.cat([model(x[:2]), model(x[2:])]).clone()
    quarters = torch.cat([model(x[i:i + 1]) for i in range(4)]).clone()
    builds = eng.plan_builds
    try:
        eng.lanes = 2
        assert eng.lanes_for(4) == 2 and eng.lanes_for(3) == 1 and eng.lanes_for(1) == 1
        got = model(x)
        assert torch.equal(got, halves)
        _check(got, want, "two lanes vs oracle")
        assert (got - single).abs().max().item() <= 1e-4 and torch.equal(got.argmax(1), single.argmax(1))
        assert eng.plan_builds == builds + 1                       # lane 0 reuses the 2-clip plan, lane 1 owns a second one
        plans = eng.lane_plans(model, x)
        assert len(plans) == 2 and plans[0] is not plans[1] and plans[0].shape == plans[1].shape == (2, 3, 8, 64, 64)
        assert len({a.t.data_ptr() for p_ in plans for a in p_.acts}) == sum(len(p_.acts) for p_ in plans)   # no shared buffer
        for _ in range(5):                                         # back-to-back calls: the lanes' buffers are reused safely
            assert torch.equal(model(x), halves)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            on_side = model(x)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(on_side, halves)
        odd = model(x[:3])                                         # 3 clips: the single-plan path, no error
        assert odd.shape == (3, 17)
        eng.lanes = 1
        assert torch.equal(odd, model(x[:3]))
        eng.lanes = 4
        assert torch.equal(model(x), quarters)
        eng.lanes = 2
        with torch.no_grad():                                      # a weight update reaches BOTH lanes' packed filters
            model.layer2[0].bn1.weight.mul_(1.5)
        upd = model(x)
        eng.lanes = 1
        assert not torch.equal(upd, halves) and torch.equal(upd, torch.cat([model(x[:2]), model(x[2:])]))
        eng.lanes = 2
        eng.use_graph = True
        assert eng.lanes_for(4) == 1
        eng.use_graph = False
        with pytest.raises(ptx.PtxError):
            eng.lanes = 0
        with pytest.raises(ptx.PtxError):
            eng.lanes = 2.0
    finally:
        eng.use_graph = False
        eng.lanes = 1
    # the non-local composite: attention is per clip, so a lane changes nothing but the batch its clips arrive in
    nl, sd2 = _build(ptx, "nonlocal_r2plus1d50", dict(num_classes=11), 5, inner_bn_damp=0.9, nl_bn_damp=0.05)
    xn = synth_clips(2, 8, 56, 3).to(DEV)
    a = nl(xn).clone()
    nl.engine().lanes = 2
    try:
        b = nl(xn)
    finally:
        nl.engine().lanes = 1
    sl = torch.cat([nl(xn[:1]), nl(xn[1:])])
    assert torch.equal(b, sl), "lanes vs slice forwards: max|d| %.3e" % (b - sl).abs().max().item()
    scale = max(1.0, a.abs().max().item())
    assert (a - b).abs().max().item() <= 1e-4 * scale, "lanes vs single plan: %.3e (scale %.2f)" % ((a - b).abs().max().item(), scale)


def test_autotune_keeps_parity(ptx):
    blob = load_golden("resnet3d50_small")
    model, _ = _build(ptx, "resnet3d50", dict(num_classes=339, pretrained=None), int(blob["w_seed"]))
    x = golden_input(blob).to(DEV)
    before = model(x).cpu()
    model.engine().autotune(model, x, iters=1)
    after = model(x).cpu()
    ref = torch.from_numpy(blob["logits"])
    _check(after, ref, "autotuned logits vs golden")
    assert (before - after).abs().max().item() <= 1e-4


def test_trn_relation_heads(ptx):
    blob = load_golden("trn_relation")
    g = torch.Generator().manual_seed(int(blob["x_seed"]))
    x = torch.randn(4, 1, 8, 256, generator=g)
    rel = ptx.Relation(8, 256, 96, 128)
    rel.load_state_dict(synth_state_dict(rel.state_dict(), int(blob["w_seed"])))
    rel = rel.to(DEV)
    _check(rel(x.to(DEV)), torch.from_numpy(blob["relation"]), "Relation vs golden", 1e-4)
    msr = ptx.MultiScaleRelation(8, 256, 96, 128, 3)
    msr.load_state_dict(synth_state_dict(msr.state_dict(), int(blob["w_seed"])))
    msr = msr.to(DEV)
    np.random.seed(int(blob["np_seed"]))          # the reference samples subsets from numpy's global RNG
    _check(msr(x.to(DEV)), torch.from_numpy(blob["multiscale"]), "MultiScaleRelation vs golden", 1e-4)
    # B == 1 still returns [1, 1, out] here (the reference's TRN.squeeze() quirk lives in TRN, not Relation)
    assert tuple(rel(x[:1].to(DEV)).shape) == (1, 1, 96)


def test_clip_parallel_world1_on_gpu(ptx):
    """world-size-1 RCCL communicator on the leased GPU: the gather path is a no-op copy."""
    import os
    import torch.distributed as dist
    from pretorched_x_amd.parallel import clip_parallel_forward, shard_clips
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        model, _ = _build(ptx, "resnet3d10", dict(), 1)
        x = synth_clips(3, 4, 32, 4).to(DEV)
        out = clip_parallel_forward(model, shard_clips(x), total=3)
        assert torch.equal(out, model(x))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", TRN_CASES)
def test_trn_wrapper_parity(ptx, case):
    """TRN.features / logits / forward (trn.py:246-263) on the GPU: per-frame 2-D resnet50 through the
    HIP engine, relation MLPs and classifier through ptx_linear_fwd -- against the real reference's
    outputs (golden; its backbone arithmetic is the torchvision stand-in) and the oracle."""
    kw, model, x, blob = golden_trn(ptx, case)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).eval()
    xd = x.to(DEV)
    seed = int(blob["np_seed"])
    if seed >= 0:
        np.random.seed(seed)
    feats = model.features(xd)
    logits = model.logits(feats)
    if seed >= 0:
        np.random.seed(seed)
    fwd = model(xd)
    torch.cuda.synchronize()
    assert feats.is_cuda and logits.is_cuda
    assert torch.equal(fwd, logits)
    gf = torch.from_numpy(blob["features"])
    _check(feats, gf, case + " features vs golden", _ftol(gf))
    _check(logits, torch.from_numpy(blob["logits"]), case + " logits vs golden")
    rng = np.random.RandomState(seed) if seed >= 0 else np.random
    want = OF.trn_forward(OF.ARCHS["resnet50"], sd, x, kw["num_segments"], kw["consensus"], rng)
    _check(logits, want, case + " logits vs oracle")
    if x.shape[0] > 1:
        assert torch.equal(logits.cpu().argmax(-1), want.argmax(-1))


def test_trn_user_head_and_errors(ptx):
    kw, model, x, _ = golden_trn(ptx, "trn_trn_b1")
    model = model.to(DEV).eval()
    model.last_linear = torch.nn.Identity()                 # README "last_linear" contract
    out = model(x.to(DEV))
    assert tuple(out.shape) == (kw["video_feature_dim"],)   # B == 1: squeeze() drops the batch axis too
    with pytest.raises(Exception):
        model(x)                                            # CPU tensor: no fallback
    with pytest.raises(ValueError):
        ptx.TRN(5, consensus="nope", pretrained=None)


@pytest.mark.parametrize("case", SLOWFAST_CASES)
def test_slowfast_parity(ptx, case):
    """SlowFast / SlowOnly / FastOnly (slowfast.py) on the GPU against the real reference's logits
    (golden) and, for the small cases, the oracle in the same run.  Frame subsampling is a stride of
    the fold kernel; the lateral concats are channel-slice writes."""
    model, sd, x, blob, (block, layers, mode) = golden_slowfast(ptx, case)
    model = model.to(DEV).eval()
    out = model(x.to(DEV))
    torch.cuda.synchronize()
    want = torch.from_numpy(blob["logits"])
    err = _check(out, want, case + " logits vs golden")
    assert torch.equal(out.cpu().argmax(1), want.argmax(1))
    if not case.endswith("_full"):
        _check(out, OF.slowfast_forward(sd, x, block, layers, mode), case + " logits vs oracle")
    again = model(x.to(DEV))
    assert torch.equal(out, again)                       # deterministic
    print("%s max|dlogits| = %.3e (max|logit| %.2f)" % (case, err, want.abs().max().item()))


def test_slowfast_errors(ptx):
    # the lateral convs stride time by 8: with slow_stride != 8 * fast_stride the lateral features do not
    # line up with the slow pathway (torch.cat fails upstream)
    m = ptx.slowfast.resnet50(num_classes=5, slow_stride=8).to(DEV)
    with pytest.raises(Exception):
        m(torch.zeros(1, 3, 32, 64, 64, device=DEV))
    with pytest.raises(TypeError):
        ptx.slowfast.resnet50(mode="nope")


def test_forward_frames_fused_preprocessing(ptx):
    """uint8 frames -> logits with TransformImage's tensor half (transforms/utils.py:72-75) on the device == the same
    model on the CPU-normalised fp32 clip; both run the direct stem kernels."""
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (2, 8, 64, 64, 3), dtype=torch.uint8, generator=g)
    opts = ptx.pretrained_settings["resnet3d50"]["moments"]
    clip = OF.transform_frames(frames, opts["mean"], opts["std"], opts["input_space"], opts["input_range"])
    model, sd = _build(ptx, "resnet3d50", dict(num_classes=339, pretrained=None), 1234)
    want = OF.forward(OF.ARCHS["resnet3d50"], sd, clip)
    a = model(clip.to(DEV))
    b = model.forward_frames(frames.to(DEV), opts)
    _check(a, want, "fp32 clip vs oracle")
    _check(b, want, "uint8 frames vs oracle")
    # round 3: uint8 frames are normalised on the device with the CPU ops' own fp32 operations (ptx_frames_u8_to_ncdhw:
    # bit-identical tensor) and then run the SAME direct stem kernel as an fp32 clip -> bit-identical logits
    assert torch.equal(a, b)
    plans = list(model.engine()._plans.values())
    assert [getattr(p, "stem_steps", 0) for p in plans] == [1, 1]
    assert not any(getattr(s, "label", "") == "fold_kw" for p in plans for s in p.steps)
    os.environ["PTX_STEM_DIRECT_U8"] = "0"          # the round-1 path (normalise + kW fold in one pass) stays selectable
    try:
        model.engine().invalidate()
        bf = model.forward_frames(frames.to(DEV), opts)
    finally:
        del os.environ["PTX_STEM_DIRECT_U8"]
        model.engine().invalidate()
    _check(bf, want, "uint8 frames, folded stem vs oracle")
    assert (a - bf).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
    # split-operand plans: both inputs take the direct patch-resident split stem (uint8 frames after the normalising pass)
    model.engine().precision = "x3"
    a3 = model(clip.to(DEV))
    b3 = model.forward_frames(frames.to(DEV), opts)
    _check(a3, want, "x3 fp32 clip vs oracle")
    _check(b3, want, "x3 uint8 frames vs oracle")
    assert torch.equal(a3, b3)
    plans = list(model.engine()._plans.values())
    assert sorted(getattr(p, "stem_steps", 0) for p in plans) == [1, 1]
    model.engine().precision = "fp32"
    # widths that are not multiples of 4: rows copied to a zero-padded 16-byte pitch, then the same direct stem
    g2 = torch.Generator().manual_seed(6)
    xo = torch.randn(2, 3, 8, 60, 66, generator=g2)
    wo = OF.forward(OF.ARCHS["resnet3d50"], sd, xo)
    yo = model(xo.to(DEV))
    _check(yo, wo, "W % 4 != 0 through the padded-pitch direct stem")
    po = [p for p in model.engine()._plans.values() if p.shape == (2, 3, 8, 60, 66)]
    assert len(po) == 1 and getattr(po[0], "stem_steps", 0) == 1 and any(getattr(s, "label", "") == "pad_rows" for s in po[0].steps)
    fo = torch.randint(0, 256, (2, 8, 60, 66, 3), dtype=torch.uint8, generator=g2)
    co = OF.transform_frames(fo, opts["mean"], opts["std"], opts["input_space"], opts["input_range"])
    _check(model.forward_frames(fo.to(DEV), opts), OF.forward(OF.ARCHS["resnet3d50"], sd, co), "uint8 frames of odd width")
    with pytest.raises(Exception):
        model.forward_frames(frames.to(DEV))            # pretrained=None models carry no mean/std
    with pytest.raises(Exception):
        model.forward_frames(frames, opts)              # CPU tensor
    # BGR / 0-255 settings and a two-pathway model (both stems read the same uint8 frames at their own stride)
    opts2 = dict(mean=[104.0, 117.0, 123.0], std=[58.0, 57.0, 57.5], input_space="BGR", input_range=[0, 255])
    frames2 = torch.randint(0, 256, (2, 32, 64, 64, 3), dtype=torch.uint8, generator=g)
    sf, sd2, _, _, (block, layers, mode) = golden_slowfast(ptx, "slowfast50_sf_small")
    sf = sf.to(DEV).eval()
    clip2 = OF.transform_frames(frames2, **opts2)
    _check(sf.forward_frames(frames2.to(DEV), opts2), OF.slowfast_forward(sd2, clip2, block, layers, mode),
           "slowfast uint8 frames vs oracle")
    # 2-D model: [N,H,W,3]
    m2, sdr = _build(ptx, "resnet18", dict(num_classes=10, pretrained=None), 7)
    img = torch.randint(0, 256, (3, 96, 80, 3), dtype=torch.uint8, generator=g)
    clip3 = OF.transform_frames(img.unsqueeze(1), opts["mean"], opts["std"])[:, :, 0]
    _check(m2.forward_frames(img.to(DEV), opts), OF.forward(OF.ARCHS["resnet18"], sdr, clip3), "2-D uint8 frames")


@pytest.mark.parametrize("shape", [(2, 3, 16, 224, 224), (2, 3, 64, 224, 224)])
def test_i3d_parity(ptx, shape):
    """I3D (BASELINE.json config 4; 2 clips per GPU at 8 GPUs) against the stand-in CPU oracle
    (**parity unpinned**: the reference snapshot has no I3D source)."""
    from oracle import i3d_standin as I3
    from pretorched_x_amd.testing import I3D_RECIPE
    model = ptx.i3d(400)
    sd = synth_state_dict(model.state_dict(), 1234, **I3D_RECIPE)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(99))
    out = model(x.to(DEV))
    feats = model.features(x.to(DEV))
    torch.cuda.synchronize()
    want = I3.forward(sd, x)
    err = _check(out, want, "i3d logits vs oracle")
    assert torch.equal(out.cpu().argmax(1), want.argmax(1))
    wf = I3.features(sd, x)
    _check(feats, wf, "i3d Mixed_5c vs oracle", _ftol(wf))
    print("i3d %s max|dlogits| = %.3e (max|logit| %.2f)" % (shape, err, want.abs().max().item()))
    if shape[2] == 16:
        with pytest.raises(Exception):
            model(torch.zeros(1, 3, 16, 112, 112, device=DEV))      # AvgPool3d([2,7,7]) needs a 7x7 map
        model.replace_logits(17)
        assert tuple(model(x.to(DEV)).shape) == (2, 17)


@pytest.mark.parametrize("res,ch,batch", [(256, 128, 3), (128, 32, 4), (32, 16, 5)])
def test_biggan_generator_parity(ptx, res, ch, batch):
    """BigGAN-deep generator (BASELINE.json config 5, fp32 path) against the stand-in CPU oracle
    (**parity unpinned**: the reference snapshot has no BigGAN source)."""
    from oracle import biggan_standin as BG
    from pretorched_x_amd.testing import BIGGAN_RECIPE
    G = ptx.biggan_deep(res, ch=ch)
    sd = synth_state_dict(G.state_dict(), 1234, **BIGGAN_RECIPE)
    G.load_state_dict(sd)
    G = G.to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    z = torch.randn(batch, 128, generator=g)
    lab = torch.randint(0, 1000, (batch,), generator=g)
    img = G(z.to(DEV), G.shared(lab.to(DEV)))
    torch.cuda.synchronize()
    want = BG.forward(sd, z, sd["shared.weight"][lab])
    assert tuple(img.shape) == (batch, 3, res, res)
    err = (img.cpu() - want).abs().max().item()
    assert err <= 1e-3, "max abs err %.3e" % err
    assert torch.equal(img, G(z.to(DEV), G.shared(lab.to(DEV))))     # deterministic
    print("biggan-deep-%d max|d image| = %.3e (pre-tanh range ~%.1f)" % (res, err, BG.pre_tanh(sd, z, sd["shared.weight"][lab]).abs().max().item()))
    with pytest.raises(Exception):
        G(z, G.shared(lab.to(DEV)).cpu())                            # CPU tensors: no fallback


def test_plan_cache_is_bounded(ptx):
    """One plan (and its activation buffers) per input shape, least-recently-used eviction."""
    model, _ = _build(ptx, "resnet3d10", dict(), 1)
    eng = model.engine()
    eng.max_plans = 3
    outs = {}
    for t in (4, 5, 6, 7, 4):
        x = synth_clips(1, t, 32, 3)
        outs.setdefault(t, []).append(model(x.to(DEV)).cpu())
        assert len(eng._plans) <= 3
    assert torch.equal(outs[4][0], outs[4][1])            # a re-compiled plan gives the same answer


def test_concurrent_callers_share_a_plan_safely(ptx):
    """Host threads on their own HIP streams calling one model with one input shape (a DataParallel-style
    worker pool on one device): the plan's single buffer set is serialised, every caller gets its own answer."""
    import threading
    model, sd = _build(ptx, "resnet3d18", dict(num_classes=50, pretrained=None), 4)
    xs = [synth_clips(2, 8, 64, 100 + i) for i in range(6)]
    want = [OF.forward(oracle_cfg("resnet3d18", {}), sd, x) for x in xs]
    model(xs[0].to(DEV))                                   # compile + tune once
    torch.cuda.synchronize()
    got, errs = [None] * len(xs), []

    def worker(i):
        try:
            st = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(st):
                xd = xs[i].to(DEV, non_blocking=False)
                for _ in range(5):
                    out = model(xd)
                st.synchronize()
                got[i] = out.cpu()
        except Exception as e:      # pragma: no cover
            errs.append(e)
    ths = [threading.Thread(target=worker, args=(i,)) for i in range(len(xs))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    for g, w in zip(got, want):
        _check(g, w, "concurrent caller")


def test_nonlocal_block_modes(ptx):
    """Standalone NonLocalBlock3D on the GPU: embedded_gaussian / dot_product / gaussian, with and without
    sub_sample and bn_layer, against the real reference's outputs (golden) and the oracle."""
    blob = load_golden("nlblock")
    x = golden_input(blob)
    for mode, sub, bn in [("embedded_gaussian", False, True), ("embedded_gaussian", True, True), ("dot_product", False, True),
                          ("dot_product", True, False), ("gaussian", False, True), ("gaussian", True, False),
                          ("concatenation", False, True), ("concatenation", True, False)]:
        tag = "%s_%d_%d" % (mode, sub, bn)
        blk = ptx.NonLocalBlock3D(16, mode=mode, sub_sample=sub, bn_layer=bn)
        sd = synth_state_dict(blk.state_dict(), int(blob["w_seed"]))
        blk.load_state_dict(sd)
        blk = blk.to(DEV).eval()
        y = blk(x.to(DEV))
        torch.cuda.synchronize()
        ref = torch.from_numpy(blob[tag])      # a feature map, not logits: 1e-4 of its scale (still inside the 1e-3 bar)
        _check(y, ref, "nlblock %s vs golden" % tag, min(TOL, 1e-4 * max(1.0, ref.abs().max().item())))
        with torch.no_grad():
            want = OF.nonlocal_block({"b." + k: v for k, v in sd.items()}, x, "b", mode, sub, bn)
        _check(y, want, "nlblock %s vs oracle" % tag, min(TOL, 1e-4 * max(1.0, want.abs().max().item())))
    with pytest.raises(Exception):
        ptx.NonLocalBlock3D(16, mode="gaussian", sub_sample=True).to(DEV)(torch.zeros(1, 16, 1, 4, 4, device=DEV))


def test_mnist_nonlocal_net(ptx):
    """MNISTNonLocalNet (nonlocalnet.py:273-309) on the GPU against the real reference's logits (golden) and the oracle:
    single-channel 3x3 convs with bias + BN + ReLU, MaxPool2d(2), two 2-D non-local blocks (fused attention), the
    NCHW-ordered flatten and the two-layer classifier."""
    blob = load_golden("mnist_nl")
    x = golden_input(blob)
    net = ptx.MNISTNonLocalNet()
    sd = synth_state_dict(net.state_dict(), **golden_recipe(blob))
    net.load_state_dict(sd)
    net = net.to(DEV).eval()
    y = net(x.to(DEV))
    torch.cuda.synchronize()
    ref = torch.from_numpy(blob["logits"])
    _check(y, ref, "mnist_nl vs golden", 1e-4)
    assert torch.equal(y.cpu().argmax(1), ref.argmax(1))
    _check(y, OF.mnist_nonlocal_forward(sd, x), "mnist_nl vs oracle", 1e-4)
    with pytest.raises(ptx.PtxError):
        net(torch.zeros(2, 1, 32, 32, device=DEV))


@pytest.mark.parametrize("width", [512, 1024])
def test_nonlocal_block_reference_widths(ptx, width):
    """The non-local modes at the reference's own widths (nonlocalresnet3d50: C = 512 / 1024), all on the fused attention
    kernel: `gaussian` has theta = phi = x, i.e. d = C = 1024 (theta fragments from global memory), `concatenation` is
    relu(a_i + b_j) / N as a 2-term dot product -- no [N, S, S] affinity in HBM for any of them."""
    g = torch.Generator().manual_seed(width)
    x = torch.randn(2, width, 2, 7, 7, generator=g) * 0.05       # gaussian mode: f_ii = |x_i|^2 ~ 2.6 at C = 1024
    for mode, sub, bn in [("gaussian", False, True), ("gaussian", True, False), ("concatenation", False, True),
                          ("concatenation", True, False), ("embedded_gaussian", True, True), ("dot_product", False, False)]:
        blk = ptx.NonLocalBlock3D(width, mode=mode, sub_sample=sub, bn_layer=bn)
        sd = synth_state_dict(blk.state_dict(), 77)
        blk.load_state_dict(sd)
        blk = blk.to(DEV).eval()
        y = blk(x.to(DEV))
        torch.cuda.synchronize()
        plan = list(blk.engine()._plans.values())[-1]
        labels = [getattr(s, "label", "") for s in plan.steps]
        assert "nonlocal_unfused" not in labels and "nonlocal_attention" in labels, (mode, labels)
        with torch.no_grad():
            want = OF.nonlocal_block({"b." + k: v for k, v in sd.items()}, x, "b", mode, sub, bn)
        _check(y, want, "nlblock C=%d %s sub=%d vs oracle" % (width, mode, sub), min(TOL, 1e-4 * max(1.0, want.abs().max().item())))
        assert (want - x).abs().max().item() > 1e-3          # the block does something (W is not at its zero init)


@pytest.mark.parametrize("dim", [1, 2])
def test_nonlocal_block_1d_2d(ptx, dim):
    """NonLocalBlock1D / NonLocalBlock2D (nonlocalnet.py:246-261) on the GPU -- the T = 1 (H = 1) case of the 3-D plan
    with a dimension-aware sub_sample window -- against the real reference's outputs (golden) and the oracle."""
    blob = load_golden("nlblock%dd" % dim)
    x = golden_input(blob)
    cls = {1: ptx.NonLocalBlock1D, 2: ptx.NonLocalBlock2D}[dim]
    for mode, sub, bn in [("embedded_gaussian", False, True), ("embedded_gaussian", True, True), ("dot_product", False, True),
                          ("dot_product", True, False), ("gaussian", False, True), ("gaussian", True, False),
                          ("concatenation", False, True), ("concatenation", True, False)]:
        tag = "%s_%d_%d" % (mode, sub, bn)
        blk = cls(16, mode=mode, sub_sample=sub, bn_layer=bn)
        sd = synth_state_dict(blk.state_dict(), int(blob["w_seed"]))
        blk.load_state_dict(sd)
        blk = blk.to(DEV).eval()
        y = blk(x.to(DEV))
        torch.cuda.synchronize()
        assert tuple(y.shape) == tuple(x.shape)
        ref = torch.from_numpy(blob[tag])
        _check(y, ref, "nlblock%dd %s vs golden" % (dim, tag), min(TOL, 1e-4 * max(1.0, ref.abs().max().item())))
        with torch.no_grad():
            want = OF.nonlocal_block({"b." + k: v for k, v in sd.items()}, x, "b", mode, sub, bn)
        _check(y, want, "nlblock%dd %s vs oracle" % (dim, tag), min(TOL, 1e-4 * max(1.0, want.abs().max().item())))


def test_i3d_forward_frames(ptx):
    """uint8 frames through the SAME-padded kW-folded I3D stem (the fold consumes the front pad of SAME)."""
    from oracle import i3d_standin as I3
    from pretorched_x_amd.testing import I3D_RECIPE
    model = ptx.i3d(400)
    sd = synth_state_dict(model.state_dict(), 1234, **I3D_RECIPE)
    model.load_state_dict(sd)
    model = model.to(DEV).eval()
    frames = torch.randint(0, 256, (1, 16, 224, 224, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(8))
    opts = dict(mean=[0.5, 0.5, 0.5], std=[0.5, 0.5, 0.5], input_space="RGB", input_range=[0, 1])   # I3D's [-1, 1] scaling
    clip = OF.transform_frames(frames, **opts)
    _check(model.forward_frames(frames.to(DEV), opts), I3.forward(sd, clip), "i3d uint8 frames vs oracle")


def test_biggan_generator_fp16(ptx):
    """BigGAN-deep-256 with fp16 MFMA operands (config 5's precision): fp32 accumulate / skip / output.  Against the
    fp32 stand-in oracle -- the tolerance is the builder's choice (parity unpinned): half-rounded activations
    through 49 convs."""
    from oracle import biggan_standin as BG
    from pretorched_x_amd.testing import BIGGAN_RECIPE
    G = ptx.biggan_deep(256, precision="fp16")
    sd = synth_state_dict(G.state_dict(), 1234, **BIGGAN_RECIPE)
    G.load_state_dict(sd)
    G = G.to(DEV).eval()
    g = torch.Generator().manual_seed(3)
    z, lab = torch.randn(3, 128, generator=g), torch.randint(0, 1000, (3,), generator=g)
    img = G(z.to(DEV), G.shared(lab.to(DEV)))
    torch.cuda.synchronize()
    want = BG.forward(sd, z, sd["shared.weight"][lab])
    err = (img.cpu() - want).abs()
    print("biggan-deep-256 fp16: max|d image| = %.3e, mean %.3e" % (err.max().item(), err.mean().item()))
    assert err.max().item() <= 5e-2 and err.mean().item() <= 3e-3
    assert torch.equal(img, G(z.to(DEV), G.shared(lab.to(DEV))))


def test_biggan_fp16_adversarial_bn_tables_take_the_fp32_affine(ptx, monkeypatch):
    """the output layer's BatchNorm with an adversarial table -- channel 0 at scale 7e4 (above the half range),
    every other channel at |mean| = 1e3 sigma (its fp16 FMA cancels: 2^-11 * 1e3 = half a standard deviation of error) --
    must come out at the plan's usual 5e-2 bound, because the host guard (plans.half_affine_ok) sends such a table through
    the producer's fp32 affine instead of the consumer's packed-fp16 FMAs; with the guard switched off the same weights FAIL
    (that is the point of the guard).  Weights are arranged so the exact answer stays O(1): channel 0 of the residual
    stream is identically zero (its scale multiplies 0), the other channels' mean is the last conv4's bias."""
    from oracle import biggan_standin as BG
    from pretorched_x_amd.testing import BIGGAN_RECIPE
    outs = {}
    for guard in ("1", "0"):
        monkeypatch.setenv("PTX_HALF_AFFINE_GUARD", guard)
        G = ptx.biggan_deep(128, ch=32, precision="fp16")
        sd = synth_state_dict(G.state_dict(), 1234, **BIGGAN_RECIPE)
        eps = float(G.bn_eps)
        bw2 = G.bottom_width ** 2
        for k in list(sd):
            if k.endswith((".conv4.weight", ".conv4.bias", ".o.weight", ".o.bias")):
                sd[k][0] = 0.0                                   # nothing ever writes channel 0 of the residual stream
        sd["linear.weight"][:bw2] = 0.0                          # (the first linear's rows are (c, h, w)-ordered)
        sd["linear.bias"][:bw2] = 0.0
        last = max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        sigma = (sd["output_layer.0.stored_var"] + eps).sqrt()
        sd["output_layer.0.stored_mean"] = 1e3 * sigma
        sd["blocks.%d.1.conv4.bias" % last] = sd["blocks.%d.1.conv4.bias" % last] + sd["output_layer.0.stored_mean"]
        sd["blocks.%d.1.conv4.bias" % last][0] = 0.0
        sd["output_layer.0.stored_mean"][0] = 0.0
        sd["output_layer.0.stored_var"][0] = 0.0                 # sigma_0 = sqrt(eps)
        sd["output_layer.0.gain"][0] = 7e4 * eps ** 0.5          # scale_0 = 7e4, times (x_0 - mean_0) = 0
        G.load_state_dict(sd)
        G = G.to(DEV).eval()
        g = torch.Generator().manual_seed(5)
        z, lab = torch.randn(2, 128, generator=g), torch.randint(0, 1000, (2,), generator=g)
        img = G(z.to(DEV), G.shared(lab.to(DEV)))
        torch.cuda.synchronize()
        want = BG.forward(sd, z, sd["shared.weight"][lab])
        assert torch.isfinite(want).all() and want.abs().max().item() < 1.0 and want.std().item() > 1e-2      # a real image, not saturated
        err = (img.cpu() - want).abs()
        outs[guard] = float("inf") if not torch.isfinite(err).all() else err.max().item()
        plan = list(G.engine()._plans.values())[-1]
        assert sum(1 for s_ in plan.steps if getattr(s_, "label", "") == "rgb_conv3x3") == (0 if guard == "1" else 1)
    print("biggan fp16, adversarial output-BN tables: max|d image| guarded %.3e, unguarded %.3e" % (outs["1"], outs["0"]))
    assert outs["1"] <= 5e-2, outs
    assert not outs["0"] <= 5e-2, outs


@pytest.mark.parametrize("arch,kw,shape", [("resnet3d50", dict(num_classes=17, pretrained=None), (3, 3, 8, 112, 112)),
                                           ("r2plus1d50", dict(num_classes=17), (2, 3, 16, 112, 112)),
                                           ("nonlocal_r2plus1d50", dict(num_classes=17), (2, 3, 16, 112, 112))])
@pytest.mark.parametrize("precision", ["fp32", "x3"])
def test_chained_and_unchained_plans_agree(ptx, arch, kw, shape, precision, monkeypatch):
    """Chained launches (conv -> 1x1x1 conv in one kernel: bottleneck tails, (2+1)D pointwise pairs) against the
    plan that runs every conv as its own launch, and against the CPU oracle: PTX_CHAIN_FORCE=1 / 0 pin the choice the tuner
    otherwise makes per pair.  Same arithmetic on both sides (bit-identical per tile shape; the two plans may pick different
    tiles, hence 1e-4 of the logits' scale instead of equality)."""
    from pretorched_x_amd.engine import AltStep
    x = synth_clips(shape[0], shape[2], shape[3], 31)
    outs = {}
    for force in ("1", "0"):
        monkeypatch.setenv("PTX_CHAIN_FORCE", force)
        recipe = dict(inner_bn_damp=0.9, nl_bn_damp=0.05) if "r2plus1d" in arch else {}
        model, sd = _build(ptx, arch, kw, 77, **recipe)
        model.engine().precision = precision          # "x3": the chained launches run on the split-operand chained tiles
        outs[force] = model(x.to(DEV))
        torch.cuda.synchronize()
        plan = list(model.engine()._plans.values())[-1]
        alts = [s for s in plan.steps if isinstance(s, AltStep)]
        assert alts and all(a.use_chain == (force == "1") for a in alts), (arch, force, len(alts))
        assert all(a.chain.kernel.endswith("/x3") == (precision == "x3") for a in alts), arch
        n_chain = sum(a.use_chain for a in alts)
        assert len(plan.all_convs()) == len(plan.conv_steps) + getattr(plan, "stem_steps", 0) - n_chain, arch
    want = OF.forward(oracle_cfg(arch, kw), sd, x)
    _check(outs["1"], want, "%s chained vs oracle" % arch)
    _check(outs["0"], want, "%s unchained vs oracle" % arch)
    assert (outs["1"] - outs["0"]).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
    assert torch.equal(outs["1"].argmax(1).cpu(), want.argmax(1))


def test_biggan_generator_full_batch_both_chunks(ptx):
    """BASELINE config 5 at its own size: batch 64.  Engine.generate runs it as two 32-image chunks (the 256^2 stage of a
    64-image batch exceeds the 2 GiB per-launch limit); ALL 64 images -- both chunks -- are compared with the stand-in
    oracle (the second chunk used to go unchecked).  fp16 operands, the builder's 5e-2 bound (parity
    unpinned: no BigGAN source in the reference snapshot), and every image must also equal the one a batch-32 call of
    its own chunk produces (chunking is invisible)."""
    from oracle import biggan_standin as BG
    from pretorched_x_amd.testing import BIGGAN_RECIPE
    G = ptx.biggan_deep(256, precision="fp16")
    sd = synth_state_dict(G.state_dict(), 1234, **BIGGAN_RECIPE)
    G.load_state_dict(sd)
    G = G.to(DEV).eval()
    g = torch.Generator().manual_seed(64)
    z, lab = torch.randn(64, 128, generator=g), torch.randint(0, 1000, (64,), generator=g)
    y = G.shared(lab.to(DEV))
    img = G(z.to(DEV), y)
    torch.cuda.synchronize()
    assert tuple(img.shape) == (64, 3, 256, 256)
    plans = list(G.engine()._plans.values())
    # halfs between the convs: the fp16 plan's largest tensor stays under the 2 GiB per-launch limit at batch 64 (one
    # launch set); the fp32 plan runs the batch as two 32-image chunks of ONE 32-image plan
    assert len(plans) == 1 and plans[0].shape[0] in (32, 64)
    chunked = plans[0].shape[0] == 32
    # a direct batch-32 call runs ANOTHER plan (its own buffers and tuned tiles: other fp16 rounding points), so an image
    # of the 64-batch equals the same image from a 32-batch call only up to the fp16 operand noise -- measured 1e-2 at the
    # worst pixel (scripts/gpu_biggan_chunk_probe.py: the plans themselves are deterministic and carry no state from one
    # batch to the next)
    second, first = G(z[32:].to(DEV), y[32:]), G(z[:32].to(DEV), y[:32])
    assert (img[32:] - second).abs().max().item() <= 2.5e-2 and (img[:32] - first).abs().max().item() <= 2.5e-2
    assert torch.equal(second, G(z[32:].to(DEV), y[32:]))                 # ... and each plan is bit-reproducible
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    got = img.cpu()
    worst = 0.0
    for i in range(0, 64, 8):                                            # bounded CPU memory: 8 images per oracle call
        with torch.no_grad():
            want = BG.forward(sd, z[i:i + 8], sd["shared.weight"][lab[i:i + 8]])
        err = (got[i:i + 8] - want).abs()
        worst = max(worst, err.max().item())
        assert err.max().item() <= 5e-2 and err.mean().item() <= 3e-3, (i, err.max().item(), err.mean().item())
    print("biggan-deep-256 fp16, batch 64 (%s): max|d image| over all 64 images = %.3e" % ("2 x 32" if chunked else "one launch set", worst))


def test_dataparallel_replicas_share_one_plan_per_device(ptx):
    """torch.nn.DataParallel (reference examples/imagenet_eval.py:136, nonlocalnet.py:604) rebuilds its replicas
    on every forward and runs them from worker threads.  Two replicas on the one device of this box: across 3
    forwards exactly ONE plan is compiled and the filters are packed ONCE; results equal the direct call."""
    from torch.nn.parallel import parallel_apply, replicate
    model, _ = _build(ptx, "resnet3d10", dict(), 21)
    eng = model.engine()
    x = synth_clips(4, 4, 32, 8).to(DEV)
    want = model(x)
    builds, packs = eng.plan_builds, eng.weight_refreshes
    assert (builds, packs) == (1, 1)
    for _ in range(3):
        reps = replicate(model, [0, 0])
        assert next(reps[0].parameters(), None) is None          # replicas expose no parameters()
        outs = parallel_apply(reps, [(x,), (x,)], devices=[0, 0])
        torch.cuda.synchronize()
        for o in outs:
            assert torch.equal(o, want)
    assert eng.plan_builds == builds and eng.weight_refreshes == packs
    # an owner-side weight edit reaches the replicas' next forward (their tensors are fresh broadcast copies)
    with torch.no_grad():
        model.bn1.weight.mul_(1.25)
    outs = parallel_apply(replicate(model, [0, 0]), [(x[:2],), (x[2:],)], devices=[0, 0])
    torch.cuda.synchronize()
    got = torch.cat(outs, 0)
    want2 = model(x)
    assert (want2 - want).abs().max().item() > 1e-4
    assert (got - want2).abs().max().item() <= 1e-5 * max(1.0, want2.abs().max().item())
    # torch.nn.DataParallel itself (one device: the wrapper forwards straight to the module)
    dp = torch.nn.DataParallel(model, device_ids=[0])
    assert torch.equal(dp(x), want2)


def test_weight_edits_through_data_need_refresh_or_checksum(ptx):
    """`p.data.fill_()`-style edits (the reference's own init idiom, resnet3D.py:199-201) bypass torch's version
    counters: documented to need model.refresh(); `check_weights = "checksum"` notices them on the device."""
    model, _ = _build(ptx, "resnet3d10", dict(), 5)
    x = synth_clips(2, 4, 32, 3).to(DEV)
    base = model(x).clone()
    model.bn1.weight.data.mul_(1.5)
    stale = model(x)
    assert torch.equal(stale, base)                       # not noticed (by design: O(#tensors) host check only)
    fresh = model.refresh()(x)
    assert (fresh - base).abs().max().item() > 1e-4
    model.engine().check_weights = "checksum"
    again = model(x)
    assert torch.equal(again, fresh)
    model.bn1.weight.data.mul_(1.0 / 1.5)
    back = model(x)                                        # noticed without refresh()
    assert (back - base).abs().max().item() <= 1e-5 * max(1.0, base.abs().max().item())
    model.engine().check_weights = False                   # O(1) mode: only refresh() / load_state_dict re-pack
    with torch.no_grad():
        model.bn1.weight.mul_(2.0)
    assert torch.equal(model(x), back)
    assert not torch.equal(model.refresh()(x), back)


X3_CASES = ["resnet3d50_small", "resnet3d50_odd", "resnet3d18_small", "nonlocalresnet3d50_small", "r2plus1d50_small",
            "nonlocal_r2plus1d50_small", "resnet18_cfg1", "mvresnet50_small"]


@pytest.mark.parametrize("case", X3_CASES)
def test_x3_split_precision_parity_small(ptx, case):
    """Engine.precision = "x3" (fp32 operands split into half pairs, three fp16 MFMAs, fp32 accumulate) must meet
    the SAME bar as the fp32 path against the real reference's goldens -- 1e-3, identical argmax -- and stay within
    1e-4 of the fp32-MFMA engine itself (it is an fp32-accurate mode, not an fp16 one)."""
    arch, kw = GOLDEN_CASES[case]
    blob = load_golden(case)
    model, sd = _build(ptx, arch, kw, **golden_recipe(blob))
    xd = golden_input(blob).to(DEV)
    base = model(xd).clone()
    model.engine().precision = "x3"
    out = model(xd)
    torch.cuda.synchronize()
    plan = list(model.engine()._plans.values())[-1]
    names = [ptx._lib.lib().ptx_conv3d_config_name(s.cfg).decode() for s in plan.conv_steps]
    assert plan.x3 and sum(n.endswith("/x3") for n in names) >= 0.9 * len(names), names
    ref = torch.from_numpy(blob["logits"])
    err = _check(out, ref, case + " x3 logits vs golden")
    assert torch.equal(out.cpu().argmax(1), ref.argmax(1))
    assert (out - base).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    model.engine().precision = "fp32"
    again = model(xd)
    assert torch.equal(again, base), "switching back must reproduce the fp32 path bit for bit"
    print("%s x3 max|dlogits| = %.3e" % (case, err))


def test_x3_config2_full_size_parity(ptx):
    blob = load_golden("resnet3d50_cfg2")
    model, sd = _build(ptx, "resnet3d50", dict(num_classes=339, pretrained=None), int(blob["w_seed"]))
    model.engine().precision = "x3"
    x = synth_clips(8, 16, 224, int(blob["x_seed"]))
    out = model(x.to(DEV))
    torch.cuda.synchronize()
    ref = torch.from_numpy(blob["logits"])
    err = _check(out, ref, "cfg2 x3 logits vs golden")
    assert err <= 2e-4, err            # fp32-class: the fp32-MFMA path measures 1.5e-5 here
    assert torch.equal(out.cpu().argmax(1), ref.argmax(1))
    print("cfg2 x3 max|dlogits| = %.3e (max|logit| %.2f)" % (err, ref.abs().max().item()))


def test_rccl_single_rank_collectives(ptx):
    """RCCL itself on the box: a 1-rank "nccl" process group (the only size a 1-GPU box can form) runs the collectives of
    the clip-parallel path on device tensors -- all_gather_into_tensor, the all_reduce self-check of
    parallel.verify_gather and the tuned-table broadcast -- in a fresh process (process groups are per process)."""
    import subprocess
    import sys
    code = r'''
import os, sys, importlib, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ptx = importlib.import_module("pretorched_x_amd")
from pretorched_x_amd.parallel import verify_gather, broadcast_tuned_table
x = torch.randn(8, 339, device="cuda")
out = x.new_empty((8, 339))
dist.all_gather_into_tensor(out, x)
assert torch.equal(out, x)
v = verify_gather(x, out)
assert v == {"gather_order_ok": True, "replicas_identical": True, "ranks": 1, "rows": 8}, v
assert broadcast_tuned_table(src=0) > 0
dist.barrier()
dist.destroy_process_group()
print("rccl-ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


@pytest.mark.parametrize("case", ["nonlocal_r2plus1d50_cfg3_fullnl", "nonlocalresnet3d50_16x224_fullnl"])
def test_nl_batch_size_and_stream_equivalence(ptx, case):
    """the reference is batch-independent by construction (per-sample softmax, eval-mode BN:
    nonlocalnet.py:143-166, :397-420) -- so must every execution shape of the HIP engine be.  The NL + (2+1)D composite at
    config 3's size (N = 1568 / 196 keys) and NonLocalResNet3D-50 at the reference's 16 x 224 x 224 input (N = 3136 / 392),
    NL branch at FULL strength (W.1 gamma undamped; the fixture's theta / phi embeddings are scaled so the softmax input
    stays in the O(1-60) range of a trained network -- tests/golden/make_golden.py RECIPES says why): one B-clip plan, two
    B/2-clip forwards sequentially on one model, B single-clip forwards, and two model instances on two HIP streams
    running concurrently, each against the REAL reference's logits (golden) and the CPU oracle (2 clips, same run).
    Bar: 3e-5 of the logits' scale (the network's fp32 noise floor on the CPU is 6e-7 ... 9e-7 of it, recorded in the
    fixture) and identical argmax."""
    arch, kw = GOLDEN_CASES[case]
    blob = load_golden(case)
    rec = golden_recipe(blob)
    model, sd = _build(ptx, arch, kw, **rec)
    x = golden_input(blob)
    xd = x.to(DEV)
    ref = torch.from_numpy(blob["logits"])
    B, h = x.shape[0], x.shape[0] // 2
    bar = min(TOL, 3e-5 * ref.abs().max().item())
    outs = {"B=%d" % B: model(xd)}
    outs["%d+%d sequential" % (h, h)] = torch.cat([model(xd[:h].contiguous()), model(xd[h:].contiguous())], 0)
    if B > 2:
        outs["1 x %d" % B] = torch.cat([model(xd[i:i + 1].contiguous()) for i in range(B)], 0)
    halves = [_build(ptx, arch, kw, **rec)[0] for _ in range(2)]
    parts = [xd[:h], xd[h:]]
    for m_, xi in zip(halves, parts):
        m_(xi)                                            # compile + tune on the default stream
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    cur = torch.cuda.current_stream()
    for rep in range(3):                                  # concurrent: both plans' launches interleave on the device
        res = []
        for m_, xi, st in zip(halves, parts, streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                res.append(m_(xi))
        for st in streams:
            cur.wait_stream(st)
    torch.cuda.synchronize()
    outs["%d+%d two instances, two streams" % (h, h)] = torch.cat(res, 0)
    worst = 0.0
    for name, o in outs.items():
        e = _check(o, ref, "%s %s vs golden" % (case, name), bar)
        worst = max(worst, e)
        assert torch.equal(o.cpu().argmax(1), ref.argmax(1)), name
    first = outs["B=%d" % B]
    for name, o in outs.items():
        assert (o - first).abs().max().item() <= bar, "%s: %s differs from the one-plan forward" % (case, name)
    want = OF.forward(oracle_cfg(arch, kw), sd, x[:2])
    _check(first[:2], want, case + " vs oracle (2 clips)", bar)
    print("%s: worst max|dlogits| over %d execution shapes = %.3e (bar %.2e, max|logit| %.2f, fixture's fp32 noise floor %.1e)" % (
        case, len(outs), worst, bar, ref.abs().max().item(), float(blob["fp32_noise_floor"])))


def test_bench_line_launch_times_sum_to_the_step(ptx):
    """the per-launch times of the bench line must be consistent with the step they are part of -- the launches
    of a step cannot sum to more than the step (round 5's rows summed to 5.76 ms in a 5.39 ms step because every launch was
    bracketed by its own event pair and the marker overhead stayed in the rows).  Engine.profile_steps now chains the events,
    times plain passes in the same call and removes the per-launch overhead; bench.py states the invariant in the line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3", "--no-autotune",
                        "--no-cpu-baseline", "--no-x3", "--no-lanes", "--lanes", "1"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    lt = line["launch_timing"]
    total = line["roofline_net"]["conv_ms_sum"] + line["non_conv_ms"]
    print("bench line: launches sum %.4f ms, plain pass %.4f ms, step %.4f ms, marker overhead %.2f us / launch (%d launches)" % (
        total, lt["plain_pass_ms"], line["ms_per_step"], lt["overhead_us_per_launch"], lt["launches"]))
    assert lt["launch_sum_le_step"] and total <= line["ms_per_step"] * 1.01, (total, line["ms_per_step"])
    assert abs(total - lt["plain_pass_ms"]) <= 0.01 * lt["plain_pass_ms"]          # the rows ARE an un-instrumented pass
    assert lt["instrumented_pass_ms"] >= lt["plain_pass_ms"] * 0.98 and lt["clamped"] == 0
    roof = line["roofline_longest_launch"]
    assert roof["kernel"].startswith("conv_stem_f32") and 0.5 < roof["frac"] < 1.0
    # the committed rocprofv3 summary of the same command names the same kernel; the two clocks agree within a few per cent
    # (profiled passes clock lower: MI355X_MICROARCH.md, DVFS note)
    if roof["rocprof"] is not None and roof["rocprof"].get("hip_event_over_rocprof") is not None:
        assert 0.90 <= roof["rocprof"]["hip_event_over_rocprof"] <= 1.06, roof["rocprof"]
