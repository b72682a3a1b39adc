"""TEST INFRASTRUCTURE ONLY -- import the real reference (read-only tree at /root/reference).

Only `tests/`, `tests/golden/make_golden.py`, `__graft_entry__.smoke()` and bench.py's
`cpu_baseline` leg may import anything under `oracle/`.  The product package never does.

The reference (`pretorched/__init__.py:1-83`) imports torchvision / munch / torchaudio, none of
which exist in this image; SURVEY.md F5/F6 describe the stub recipe restated here.  The reference
tree does not exist on the GPU box, so everything that uses this module must be skipped there
(`have_reference()`), and the committed fixtures under tests/golden/ carry its outputs instead.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PTX_REFERENCE_ROOT", "/root/reference")


def have_reference():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "pretorched"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference(torchvision_resnets=None):
    """Return the imported reference package `pretorched` (CPU only, no bytecode written).

    torchvision_resnets: optional dict name->factory used as `torchvision.models.<name>`
    (the reference's 2-D resnet18 wrapper calls `models.resnet18`, torchvision_models.py:487;
    torchvision itself is a third-party, unpinned dependency absent from this image).
    """
    if not have_reference():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    if "pretorched" in sys.modules and getattr(sys.modules["pretorched"], "_ptx_shimmed", False):
        ref = sys.modules["pretorched"]
    else:
        if REFERENCE_ROOT not in sys.path:
            sys.path.insert(0, REFERENCE_ROOT)
        tv = _stub("torchvision")
        tv.models = _stub("torchvision.models")
        tv.transforms = _stub("torchvision.transforms")
        tv.datasets = _stub("torchvision.datasets")
        _stub("munch", munchify=lambda d: d)
        _stub("torchaudio")
        import pretorched as ref  # noqa: E402
        ref._ptx_shimmed = True
    if torchvision_resnets:
        for k, v in torchvision_resnets.items():
            setattr(sys.modules["torchvision.models"], k, v)
    return ref


def import_r2plus1d():
    """r2plus1d.py:10 uses an absolute `import resnet3D`."""
    ref = import_reference()
    import importlib
    sys.modules.setdefault("resnet3D", ref.models.resnet3D)
    return importlib.import_module("pretorched.models.r2plus1d")


def import_preact():
    """pre_act_resnet3D.py:8 uses an absolute `import resnet3D`."""
    ref = import_reference()
    import importlib
    sys.modules.setdefault("resnet3D", ref.models.resnet3D)
    return importlib.import_module("pretorched.models.pre_act_resnet3D")


def import_multiview():
    """multiview.py:10 uses an absolute `import resnet3D`."""
    ref = import_reference()
    import importlib
    sys.modules.setdefault("resnet3D", ref.models.resnet3D)
    return importlib.import_module("pretorched.models.multiview")


def import_wideresnet3d():
    """wideresnet3D.py:9 does `from torchvision_models import ...` (absolute)."""
    ref = import_reference()
    import importlib
    sys.modules.setdefault("torchvision_models", ref.models.torchvision_models)
    return importlib.import_module("pretorched.models.wideresnet3D")


def import_trn():
    """trn.py:8 does `import pretrainedmodels` (the upstream name of this package)."""
    ref = import_reference()
    import importlib
    sys.modules.setdefault("pretrainedmodels", ref)
    return importlib.import_module("pretorched.models.trn")
