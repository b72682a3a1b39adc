"""CPU tests of the oracle: (a) bit-equality with the imported reference where /root/reference
exists (builder container only), (b) agreement with the committed golden fixtures everywhere."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, golden_input, load_golden
from oracle import functional as OF
from oracle import ref_shim, tv_standin
from pretorched_x_amd.testing import synth_state_dict

needs_ref = pytest.mark.skipif(not ref_shim.have_reference(), reason="reference tree not present (GPU box)")

# CPU results move by ~1e-5 between machines / thread counts (oneDNN blocking); goldens were
# produced on the 8-core builder container
GOLDEN_TOL = 2e-4


def _arch_sd(ptx, arch, kw, seed):
    model = ptx.__dict__[arch](**kw)
    return model.arch, synth_state_dict(model.state_dict(), seed)


from conftest import FULL_SIZE


@pytest.mark.parametrize("case", [c for c in GOLDEN_CASES if c not in FULL_SIZE])
def test_oracle_matches_golden(ptx, case):
    arch, kw = GOLDEN_CASES[case]
    blob = load_golden(case)
    cfg = OF.ARCHS[arch]
    _, sd = _arch_sd(ptx, arch, kw, int(blob["w_seed"]))
    x = golden_input(blob)
    with torch.no_grad():
        feat = OF.features(cfg, sd, x)
        out = OF.logits(cfg, sd, feat)
    ref = torch.from_numpy(blob["logits"])
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() <= GOLDEN_TOL * max(1.0, ref.abs().max().item())
    assert torch.equal(out.argmax(1), ref.argmax(1))
    if "features" in blob.files:
        fr = torch.from_numpy(blob["features"])
        assert feat.shape == fr.shape
        assert (feat - fr).abs().max().item() <= GOLDEN_TOL * max(1.0, fr.abs().max().item())


def test_oracle_trn_golden():
    blob = load_golden("trn_relation")
    g = torch.Generator().manual_seed(int(blob["x_seed"]))
    x = torch.randn(4, 1, 8, 256, generator=g)
    import pretorched_x_amd as ptx
    rel = ptx.Relation(8, 256, 96, 128)
    sd = synth_state_dict(rel.state_dict(), int(blob["w_seed"]))
    y = OF.relation(sd, x, "", 8)
    assert np.abs(y.numpy() - blob["relation"]).max() <= GOLDEN_TOL
    msr = ptx.MultiScaleRelation(8, 256, 96, 128, 3)
    sd = synth_state_dict(msr.state_dict(), int(blob["w_seed"]))
    rng = np.random.RandomState(int(blob["np_seed"]))   # same stream as np.random.seed(7)
    y = OF.multiscale_relation(sd, x, 8, 3, rng)
    assert np.abs(y.numpy() - blob["multiscale"]).max() <= GOLDEN_TOL


@needs_ref
@pytest.mark.parametrize("arch,kw,shape", [
    ("resnet3d50", dict(num_classes=339, pretrained=None), (1, 3, 8, 64, 64)),
    ("resnet3d18", dict(num_classes=400, pretrained=None), (2, 3, 4, 48, 48)),
    ("resnet3d10", dict(), (1, 3, 4, 32, 32)),
    ("nonlocalresnet3d50", dict(pretrained=None), (1, 3, 8, 64, 64)),
])
def test_oracle_bit_equal_to_reference(arch, kw, shape):
    ref = ref_shim.import_reference({"resnet18": tv_standin.resnet18})
    model = ref.__dict__[arch](**kw).eval()
    sd = synth_state_dict(model.state_dict(), 5)
    model.load_state_dict(sd)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want_f = model.features(x)
        want = model.logits(want_f)
        got_f = OF.features(OF.ARCHS[arch], sd, x)
        got = OF.logits(OF.ARCHS[arch], sd, got_f)
    assert torch.equal(got_f, want_f)
    assert torch.equal(got, want)


@needs_ref
def test_oracle_bit_equal_r2plus1d_and_relation():
    r2 = ref_shim.import_r2plus1d()
    stc = r2.SpatioTemporalConv(16, 24, 3, stride=(2, 2, 2), padding=1, bias=False).eval()
    sd = synth_state_dict(stc.state_dict(), 11)
    stc.load_state_dict(sd)
    x = torch.randn(2, 16, 6, 20, 20, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        want = stc(x)
        got = OF._st_conv({"c." + k: v for k, v in sd.items()}, x, "c", (2, 2, 2), (1, 1, 1))
    assert torch.equal(got, want)
    assert stc.spatial_conv.out_channels == OF.st_mid_channels(16, 24, 3)
    trn = ref_shim.import_trn()
    rel = trn.Relation(4, 32, 10, 16).eval()
    sd = synth_state_dict(rel.state_dict(), 2)
    rel.load_state_dict(sd)
    xr = torch.randn(3, 1, 4, 32, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        assert torch.equal(OF.relation(sd, xr, "", 4), rel(xr))


@needs_ref
def test_oracle_nonlocal_modes_bit_equal():
    ref = ref_shim.import_reference()
    nl = ref.models.nonlocalnet
    x = torch.randn(2, 8, 2, 6, 6, generator=torch.Generator().manual_seed(8))
    for mode in ("embedded_gaussian", "gaussian", "dot_product", "concatenation"):
        blk = nl.NonLocalBlock3D(8, mode=mode).eval()
        sd = synth_state_dict(blk.state_dict(), 21)
        blk.load_state_dict(sd)
        with torch.no_grad():
            want = blk(x)
            got = OF.nonlocal_block({"b." + k: v for k, v in sd.items()}, x, "b", mode)
        assert torch.equal(got, want), mode


@needs_ref
def test_resnet18_standin_through_reference_wrapper():
    """config 1: reference wrapper (modify_resnets) over the torchvision stand-in == oracle 2-D path.
    Parity for this case is *unpinned* (torchvision is third-party and absent)."""
    ref = ref_shim.import_reference({"resnet18": tv_standin.resnet18})
    model = ref.resnet18(num_classes=1000, pretrained=None).eval()
    sd = synth_state_dict(model.state_dict(), 5)
    model.load_state_dict(sd)
    x = torch.randn(1, 3, 96, 96, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        assert torch.equal(OF.forward(OF.ARCHS["resnet18"], sd, x), model(x))
