"""Drop-in boundary: `accelerate(model)` for an instance the REFERENCE built, the exported-but-broken
non-local factories, `utils.Identity`.  The `needs_ref` tests import /root/reference (builder container only); the GPU
test rebuilds the same module tree from the committed golden recipe through oracle/ (test infrastructure)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import GOLDEN_CASES, golden_input, golden_recipe, load_golden
from oracle import ref_shim
from pretorched_x_amd.testing import synth_state_dict

needs_ref = pytest.mark.skipif(not ref_shim.have_reference(), reason="reference tree not present (GPU box)")


def _ref_models():
    """(name, reference instance) for the three families `accelerate` covers, built LAZILY: `modify_resnets` patches
    features / logits / forward onto the ResNet3D CLASS (torchvision_models.py:443-481), which breaks every
    R2Plus1D instance in the process (it inherits the patched forward but keeps `fc`) -- so the (2+1)D case runs first and
    is skipped when an earlier test already triggered the patch."""
    ref = ref_shim.import_reference()
    r2 = ref_shim.import_r2plus1d()
    if "logits" not in ref.models.resnet3D.ResNet3D.__dict__:
        yield "r2plus1d18", r2.r2plus1d18(num_classes=11)
    nl = ref.models.nonlocalnet
    yield "nonlocalresnet3d50", nl.nonlocalresnet3d50(pretrained=None)
    yield "resnet3d50", ref.resnet3d50(num_classes=17, pretrained=None)
    yield "resnet3d18", ref.resnet3d18(num_classes=13, pretrained=None)


@needs_ref
def test_accelerate_reference_instances_share_parameters_and_keep_the_reference_path(ptx):
    for name, m in _ref_models():
        m.eval()
        cls = type(m)
        had = {n: cls.__dict__.get(n) for n in ("features", "logits", "forward")}
        x = torch.randn(1, 3, 4, 32, 32, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            want = m(x).clone()
        got_m = ptx.accelerate(m)
        assert got_m is m
        twin = m._ptx_twin
        # state_dict identity: the same keys, the SAME tensors (no copy).  The classifier is not a child of the twin: the
        # engine reads `last_linear` / `fc` from the instance at call time (users replace it, README "last_linear")
        head = twin.arch.head + "."
        sd_ref, sd_twin = m.state_dict(keep_vars=True), twin.state_dict(keep_vars=True)
        for k, v in sd_ref.items():
            assert k.startswith(head) or (k in sd_twin and sd_twin[k] is v), (name, k)
        assert set(sd_twin) == {k for k in sd_ref if not k.startswith(head)}, (name, set(sd_twin) ^ set(sd_ref))
        assert twin.head_module is getattr(m, twin.arch.head)
        # bound on the INSTANCE: the class is untouched, a second instance of it knows nothing of the engine
        assert {n: cls.__dict__.get(n) for n in ("features", "logits", "forward")} == had
        assert "forward" in m.__dict__ and callable(m.engine)
        # CPU tensors are outside the engine's contract: the reference's own code answers, bit for bit
        with torch.no_grad():
            assert torch.equal(m(x), want), name
            if hasattr(cls, "features"):
                assert torch.equal(m.logits(m.features(x)), want), name
        # the derived architecture equals the product's own for that name
        own = ptx.ARCHS[name]
        a = twin.arch
        assert (a.block, tuple(a.layers), a.shortcut, a.conv, a.head) == (own.block, tuple(own.layers), own.shortcut, own.conv, own.head), name
        nl_ref = [hasattr(b, "nonlocalblock") for i in range(1, 5) for b in getattr(m, "layer%d" % i)]
        nl_twin = [b.has_nl for i in range(1, 5) for b in getattr(twin, "layer%d" % i)]
        assert nl_ref == nl_twin
        # and a plan compiles from the adopted tree (meta device: descriptors, tiles, buffer shapes; nothing launched)
        plan = m.engine().dry_plan(twin, (2, 3, 8, 64, 64))
        ref_plan_model = ptx.__dict__[name](**({"pretrained": None} if name == "nonlocalresnet3d50" else
                                               {"num_classes": 7} if name.startswith("r2") else {"num_classes": 7, "pretrained": None}))
        plan_own = ref_plan_model.engine().dry_plan(ref_plan_model, (2, 3, 8, 64, 64))
        assert len(plan.all_convs()) == len(plan_own.all_convs()), name
        # replacing last_linear ON THE INSTANCE is what the engine's head sees
        if a.head == "last_linear":
            m.last_linear = ptx.utils.Identity()
            assert isinstance(twin.head_module, ptx.utils.Identity)


@needs_ref
def test_accelerate_refuses_what_it_does_not_cover(ptx):
    ref = ref_shim.import_reference()
    with pytest.raises(ptx.PtxError):
        ptx.accelerate(nn.Linear(3, 3))
    with pytest.raises(ptx.PtxError):
        ptx.accelerate(ref.resnext3d50())                        # grouped bottlenecks: build through pretorched_x_amd
    own = ptx.resnet3d10()
    assert ptx.accelerate(own) is own


def test_nonlocal_factories_mirror_the_reference_signature(ptx):
    """pretorched/__init__.py:74-77 exports nonlocalresnet3d18 / 34 / 101; nonlocalnet.py:524-577 forgets the required
    `nonlocal_layers` argument, so a bare call raises TypeError.  Same here; with the argument they build."""
    for name in ("nonlocalresnet3d18", "nonlocalresnet3d34", "nonlocalresnet3d101"):
        assert name in ptx.__dict__ and name in ptx.model_names
        with pytest.raises(TypeError, match="nonlocal_layers"):
            ptx.__dict__[name]()
        with pytest.raises(TypeError, match="nonlocal_layers"):
            ptx.__dict__[name](num_classes=10)
    m = ptx.nonlocalresnet3d18(nonlocal_layers=[0, 1, 1, 0], num_classes=5)
    flags = [b.has_nl for i in range(1, 5) for b in getattr(m, "layer%d" % i)]
    assert flags == [False, False, True, False, True, False, False, False]
    assert m.arch.shortcut == "A" and m.last_linear.out_features == 5
    with torch.no_grad():
        assert m(torch.randn(1, 3, 4, 32, 32)).shape == (1, 5)
    if ref_shim.have_reference():
        ref = ref_shim.import_reference()
        with pytest.raises(TypeError, match="nonlocal_layers"):
            ref.nonlocalresnet3d18()
        rm = ref.nonlocalresnet3d18(nonlocal_layers=[0, 1, 1, 0], num_classes=5)
        assert list(rm.state_dict()) == list(m.state_dict())


def test_identity_export(ptx):
    """README.md:543-546 `model.last_linear = pretrained.utils.Identity()`; models/utils.py:81."""
    assert ptx.utils.Identity is ptx.models.Identity is ptx.models.utils.Identity
    x = torch.randn(2, 5)
    assert ptx.utils.Identity()(x) is x
    m = ptx.resnet3d10(num_classes=9)
    m.last_linear = ptx.utils.Identity()
    with torch.no_grad():
        assert m(torch.randn(1, 3, 4, 32, 32)).shape == (1, 512)


class _Bag(nn.Module):
    def forward(self, *a):
        raise RuntimeError


def _reference_shaped_resnet3d50(num_classes):
    """A module tree with the reference ResNet3D's layout and attribute names (resnet3D.py:109-218) built WITHOUT the
    reference (absent on the GPU box): nn.Sequential stages of blocks that own conv1..bn3 / downsample / stride, class-level
    features / logits / forward as `modify_resnets` leaves them (torchvision_models.py:448-469)."""
    from oracle import functional as OF        # noqa: F401  (test infrastructure: the arithmetic the class methods follow)

    class Bottleneck(nn.Module):
        def __init__(self, cin, planes, stride, down):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv3d(cin, planes, 1, bias=False), nn.BatchNorm3d(planes)
            self.conv2, self.bn2 = nn.Conv3d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm3d(planes)
            self.conv3, self.bn3 = nn.Conv3d(planes, planes * 4, 1, bias=False), nn.BatchNorm3d(planes * 4)
            self.relu = nn.ReLU(inplace=True)
            self.downsample, self.stride = down, stride

        def forward(self, x):
            r = x
            o = self.relu(self.bn1(self.conv1(x)))
            o = self.relu(self.bn2(self.conv2(o)))
            o = self.bn3(self.conv3(o))
            if self.downsample is not None:
                r = self.downsample(x)
            return self.relu(o + r)

    class ResNet3D(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv3d(3, 64, 7, (1, 2, 2), (3, 3, 3), bias=False)
            self.bn1, self.relu = nn.BatchNorm3d(64), nn.ReLU(inplace=True)
            self.maxpool = nn.MaxPool3d(3, 2, 1)
            cin = 64
            for li, (planes, n) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
                blocks = []
                for b in range(n):
                    s = 2 if (b == 0 and li > 0) else 1
                    down = None
                    if b == 0:
                        down = nn.Sequential(nn.Conv3d(cin, planes * 4, 1, s, bias=False), nn.BatchNorm3d(planes * 4))
                    blocks.append(Bottleneck(cin, planes, s, down))
                    cin = planes * 4
                setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
            self.avgpool = nn.AdaptiveAvgPool3d(1)
            self.last_linear, self.fc = nn.Linear(2048, num_classes), None

        def features(self, x):
            x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
            return self.layer4(self.layer3(self.layer2(self.layer1(x))))

        def logits(self, f):
            return self.last_linear(self.avgpool(f).view(f.size(0), -1))

        def forward(self, x):
            return self.logits(self.features(x))

    return ResNet3D()


@pytest.mark.gpu
def test_accelerate_golden_weights_through_a_foreign_instance(ptx):
    """The golden weights of resnet3d50_small loaded into a reference-shaped instance (not a pretorched_x_amd class), then
    `accelerate`: logits equal the committed reference logits at the 1e-3 bar, argmax equal; `.cuda()` AFTER accelerate and
    `last_linear` replacement on the instance are honoured."""
    arch, kw = GOLDEN_CASES["resnet3d50_small"]
    blob = load_golden("resnet3d50_small")
    own = ptx.__dict__[arch](**kw)
    sd = synth_state_dict(own.state_dict(), **golden_recipe(blob))
    m = _reference_shaped_resnet3d50(kw["num_classes"]).eval()
    m.load_state_dict(sd)
    ptx.accelerate(m)
    m.cuda()                                                     # moves the shared modules; the twin sees it
    x = golden_input(blob).cuda()
    with torch.no_grad():
        out = m(x)
        feats = m.features(x)
        out2 = m.logits(feats)
    ref = torch.from_numpy(blob["logits"])
    assert (out.cpu() - ref).abs().max().item() <= 1e-3 and torch.equal(out.cpu().argmax(1), ref.argmax(1))
    # forward() pools the channels-last feature map in the plan; features() -> logits() pools the NCDHW copy it returned:
    # the same numbers in another summation order
    assert (out - out2).abs().max().item() <= 1e-5 * max(1.0, out.abs().max().item())
    assert m.engine().plan_builds >= 1
    m.last_linear = ptx.utils.Identity()
    with torch.no_grad():
        pooled = m(x)
    assert pooled.shape == (x.shape[0], 2048)
    assert np.isfinite(pooled.cpu().numpy()).all()


@needs_ref
def test_accelerated_instances_survive_copies(ptx):
    """the bound methods of an accelerated instance resolve their twin from the instance they are called on --
    copy.deepcopy, pickle (torch.save(model)) and nn.DataParallel replicas must never run the ORIGINAL model's twin."""
    import copy
    import pickle
    ref = ref_shim.import_reference()
    m = ref.resnet3d18(num_classes=5, pretrained=None).eval()
    ptx.accelerate(m)
    x = torch.randn(1, 3, 4, 32, 32, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        want = m(x).clone()                                    # CPU tensor: the reference path answers
    for clone in (copy.deepcopy(m), pickle.loads(pickle.dumps(m))):
        assert clone is not m and clone.forward.model is clone and clone.engine.model is clone
        eng = clone.engine()                                   # first use: a twin over the CLONE's modules, its own Engine
        assert clone._ptx_twin._source is clone and eng is clone._ptx_twin._engine and eng is not m.engine()
        assert clone._ptx_twin.conv1 is clone.conv1 and clone._ptx_twin.conv1 is not m.conv1
        with torch.no_grad():
            assert torch.equal(clone(x), want)
        clone.conv1.weight.data.zero_()                        # the copies share nothing with the original
        with torch.no_grad():
            assert torch.equal(m(x), want)
    # DataParallel's replicate: __dict__ is copied, then the replica's methods are re-bound to the replica and its twin
    # shares the original's Engine (plans are per device; weight identity is the original's)
    rep = m._replicate_for_data_parallel()
    assert rep is not m and rep.forward.model is rep and rep.features.model is rep
    assert rep._ptx_twin._source is rep and rep._ptx_twin._engine is m.engine() and rep._ptx_twin._is_replica
    assert m.forward.model is m and m._ptx_twin._source is m
