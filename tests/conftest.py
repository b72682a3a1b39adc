import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible and -m gpu was not requested
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _gpu_tests_never_take_the_torch_nn_path(request):
    """`-m gpu` tests are parity tests of the HIP path: the torch.nn (train / CPU) path of eager.py must not
    serve a single forward in them."""
    if "gpu" not in request.keywords:
        yield
        return
    from pretorched_x_amd import eager
    before = eager.calls
    yield
    assert eager.calls == before, "a GPU parity test ran the torch.nn path instead of the HIP engine"


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def ptx():
    import pretorched_x_amd
    return pretorched_x_amd


# golden case name -> (arch, factory kwargs); shapes/seeds live inside the fixture files
GOLDEN_CASES = {
    "resnet3d50_small": ("resnet3d50", dict(num_classes=339, pretrained=None)),
    "resnet3d50_odd": ("resnet3d50", dict(num_classes=17, pretrained=None)),
    "resnet3d10_small": ("resnet3d10", dict()),
    "resnet3d18_small": ("resnet3d18", dict(num_classes=400, pretrained=None)),
    "resnet3d34_small": ("resnet3d34", dict(num_classes=400, pretrained=None)),
    "nonlocalresnet3d50_small": ("nonlocalresnet3d50", dict(pretrained=None)),
    "r2plus1d18_small": ("r2plus1d18", dict(num_classes=174)),
    "r2plus1d50_small": ("r2plus1d50", dict(num_classes=400)),
    "nonlocal_r2plus1d50_small": ("nonlocal_r2plus1d50", dict(num_classes=339)),
    "resnet18_cfg1": ("resnet18", dict(num_classes=1000, pretrained=None)),
    "resnet50_2d_small": ("resnet50", dict(num_classes=339, pretrained=None)),
    "resnext3d50_small": ("resnext3d50", dict(num_classes=400)),
    "resnext3d10_odd": ("resnext3d10", dict(num_classes=17)),
    "resnext3d50_full": ("resnext3d50", dict(num_classes=400)),
    "wideresnet3d50_small": ("wideresnet3d50", dict(num_classes=400, pretrained=None)),
    "preact_resnet3d50_small": ("preact_resnet3d50", dict(num_classes=339)),
    "preact_resnet3d18_odd": ("preact_resnet3d18", dict(num_classes=17, shortcut_type="A")),
    "mvresnet18_small": ("mvresnet18", dict(num_classes=339)),
    "mvresnet50_small": ("mvresnet50", dict(num_classes=174)),
    "mvresnet10_odd": ("mvresnet10", dict(num_classes=17)),
    "resnet3d50_cfg2": ("resnet3d50", dict(num_classes=339, pretrained=None)),
    "nonlocal_r2plus1d50_cfg3": ("nonlocal_r2plus1d50", dict(num_classes=339)),
    "r2plus1d50_cfg3": ("r2plus1d50", dict(num_classes=400)),
    "nonlocalresnet3d50_cfg3": ("nonlocalresnet3d50", dict(pretrained=None)),
    "nonlocal_r2plus1d50_cfg3_fullnl": ("nonlocal_r2plus1d50", dict(num_classes=339)),
    "nonlocalresnet3d50_16x224_fullnl": ("nonlocalresnet3d50", dict(pretrained=None)),
}
TRN_CASES = ("trn_htrn_small", "trn_mstrn_small", "trn_trn_b1")
SLOWFAST_CASES = ("slowfast50_sf_small", "slowfast50_s_small", "slowfast50_f_small", "slowfast18_sf_small",
                  "slowfast50_sf_full")
FULL_SIZE = ("resnet3d50_cfg2", "nonlocal_r2plus1d50_cfg3", "r2plus1d50_cfg3", "nonlocalresnet3d50_cfg3",
             "nonlocal_r2plus1d50_cfg3_fullnl", "nonlocalresnet3d50_16x224_fullnl")


def oracle_cfg(arch, kw):
    """The oracle's ArchCfg for a golden case (factory kwargs may override the shortcut type)."""
    import dataclasses
    from oracle import functional as OF
    cfg = OF.ARCHS[arch]
    return dataclasses.replace(cfg, shortcut=kw["shortcut_type"]) if "shortcut_type" in kw else cfg


def golden_recipe(blob):
    """kwargs of synth_state_dict the fixture was generated with (seed + BN damping)."""
    import json
    kw = json.loads(str(blob["recipe"])) if "recipe" in blob.files else {}
    return dict(seed=int(blob["w_seed"]), **kw)


def golden_input(blob):
    """Regenerate the fixture's input from its recorded shape and seed (CPU generator)."""
    shape = tuple(int(v) for v in blob["shape"])
    g = torch.Generator().manual_seed(int(blob["x_seed"]))
    return torch.randn(*shape, generator=g)


def golden_trn(ptx, case):
    """(model kwargs, product TRN on CPU with the fixture's synthetic weights loaded, input, blob)."""
    import json
    from pretorched_x_amd.testing import synth_state_dict
    blob = load_golden(case)
    kw = json.loads(str(blob["kwargs"]))
    model = ptx.TRN(pretrained=None, **kw)
    model.load_state_dict(synth_state_dict(model.state_dict(), int(blob["w_seed"])))
    return kw, model, golden_input(blob), blob


def golden_slowfast(ptx, case):
    """(product model on CPU with the fixture's weights, state_dict, input, blob, oracle args)."""
    import json
    from pretorched_x_amd.testing import synth_state_dict
    blob = load_golden(case)
    fac, mode, kw = str(blob["factory"]), str(blob["mode"]), json.loads(str(blob["kwargs"]))
    model = getattr(ptx.slowfast, fac)(mode=mode, **kw)
    sd = synth_state_dict(model.state_dict(), int(blob["w_seed"]))
    model.load_state_dict(sd)
    block, layers = {"resnet50": ("bottleneck", [3, 4, 6, 3]), "resnet18": ("basic", [2, 2, 2, 2])}[fac]
    return model, sd, golden_input(blob), blob, (block, layers, mode.lower())
