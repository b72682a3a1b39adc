"""pretorched-x_amd -- MI355X-native forward-pass engine behind pretorched-x's model API.

Drop-in surface (reference README.md:11-15, 137-143; pretorched/__init__.py:56-83):

    import pretorched_x_amd as pretorched            # see pretorched_x_amd.py at the repo root
    model = pretorched.__dict__['resnet3d50'](num_classes=339, pretrained=None).cuda().eval()
    feats = model.features(clips)      # [B, 2048, T/16, H/32, W/32]
    out   = model.logits(feats)        # [B, 339]
    out   = model(clips)               # == logits(features(clips))

Only the video hot path of BASELINE.json is implemented: the 3-D ResNet family, (2+1)D ResNets,
the non-local ResNet, the TRN relation heads and the 2-D resnet18 plumbing case.  Every FLOP of
`features/logits/forward` runs in libptx_amd.so (hand-written gfx950 HIP); there is no fallback.
"""
import dataclasses
from collections import defaultdict

import torch

from . import _lib
from ._lib import PtxError
from .engine import Engine, relation_mlp
from . import transforms  # noqa: F401
from . import models, utils  # noqa: F401  (utils.Identity: README.md:543-546; models.Identity: models/__init__.py:79)
from .adopt import accelerate  # noqa: F401
from .biggan import BigGANDeepGenerator, biggan_deep  # noqa: F401
from .i3d import InceptionI3d, i3d  # noqa: F401
from . import slowfast  # noqa: F401  (reference: `from .models import slowfast`, pretorched/__init__.py:83)
from .zoo import (ARCHS, TRN, Arch, HierarchicalRelation, MNISTNonLocalNet, MultiScaleHierarchicalRelation, MultiScaleRelation, MultiViewConv,
                  NonLocalBlock1D, NonLocalBlock2D, NonLocalBlock3D, Relation, VideoResNet, factored_mid_channels)

__version__ = "0.1.0"

# ---------------------------------------------------------------------------------------------
# pretrained_settings: same keys/values the reference publishes (resnet3D.py:18-55,
# nonlocalnet.py:11-47, torchvision_models.py:40-112).  URLs need network access, which this
# image does not have; with a populated $TORCH_HOME cache they load exactly as in the reference.
# ---------------------------------------------------------------------------------------------
_IMAGENET_MEAN, _IMAGENET_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
_HOST = "http://pretorched-x.csail.mit.edu/models/"
_URLS = {
    "kinetics-400": {
        "resnet3d18": _HOST + "resnet3d18_kinetics-e9f44270.pth",
        "resnet3d34": _HOST + "resnet3d34_kinetics-7fed38dd.pth",
        "resnet3d50": _HOST + "resnet3d50_kinetics-aad059c9.pth",
        "resnet3d101": _HOST + "resnet3d101_kinetics-8d4c9d63.pth",
        "resnet3d152": _HOST + "resnet3d152_kinetics-575c47e2.pth",
        "nonlocalresnet3d50": _HOST + "resnet3d50_kinetics-aad059c9.pth",
    },
    "moments": {
        "resnet3d50": _HOST + "resnet3d50_16seg_moments-6eb53860.pth",
        "resnet50": "http://moments.csail.mit.edu/moments_models/resnet50_moments-fd0c4436.pth",
    },
    "imagenet": {"resnet18": "https://download.pytorch.org/models/resnet18-5c106cde.pth",
                 "resnet34": "https://download.pytorch.org/models/resnet34-333f7ec4.pth",
                 "resnet50": "https://download.pytorch.org/models/resnet50-19c8e357.pth",
                 "resnet101": "https://download.pytorch.org/models/resnet101-5d3b4d8f.pth",
                 "resnet152": "https://download.pytorch.org/models/resnet152-b121ed2d.pth"},
    "places365": {"resnet18": _HOST + "resnet18_places365-dbad67aa.pth",
                  "resnet50": _HOST + "resnet50_places365-a570fcfc.pth"},
}
_NUM_CLASSES = {"kinetics-400": 400, "moments": 339, "imagenet": 1000, "places365": 365}

pretrained_settings = defaultdict(dict)
for _name in ["resnet3d10", "resnet3d18", "resnet3d34", "resnet3d50", "resnet3d101", "resnet3d152",
              "resnet3d200", "nonlocalresnet3d50"]:
    for _ds in ("kinetics-400", "moments"):
        pretrained_settings[_name][_ds] = {
            "input_space": "RGB", "input_range": [0, 1], "url": _URLS[_ds].get(_name),
            "std": _IMAGENET_STD, "mean": _IMAGENET_MEAN, "num_classes": _NUM_CLASSES[_ds],
            "input_size": [3, 224, 224]}
for _ds in ("imagenet", "places365"):
    for _name in _URLS[_ds]:
        pretrained_settings[_name][_ds] = {
            "input_space": "RGB", "input_range": [0, 1], "url": _URLS[_ds][_name],
            "std": _IMAGENET_STD, "mean": _IMAGENET_MEAN, "num_classes": _NUM_CLASSES[_ds],
            "input_size": [3, 224, 224]}
pretrained_settings["trn"]["moments"] = {"url": "", "num_classes": 339}       # trn.py:10-17
pretrained_settings["resnet50"]["moments"] = {
    "input_space": "RGB", "input_range": [0, 1], "url": _URLS["moments"]["resnet50"],
    "std": _IMAGENET_STD, "mean": _IMAGENET_MEAN, "num_classes": 339, "input_size": [3, 224, 224]}


def _fetch(url):
    if url is None:
        raise PtxError("no pretrained weights are published for this model/dataset pair")
    return torch.hub.load_state_dict_from_url(url, map_location="cpu")


def _apply_settings(model, settings):
    for k in ("input_space", "input_size", "input_range", "mean", "std"):
        setattr(model, k, settings[k])


def _rename_head(sd, model):
    """Checkpoints carry `fc.*`; live models call the classifier `last_linear` (torchvision_models.py:445)."""
    if model.arch.head == "last_linear":
        sd = {("last_linear." + k[3:] if k.startswith("fc.") else k): v for k, v in sd.items()}
    return sd


def load_pretrained(model, num_classes, settings):
    """reference torchvision_models.py:158-167"""
    assert num_classes == settings["num_classes"], \
        "num_classes should be {}, but is {}".format(settings["num_classes"], num_classes)
    model.load_state_dict(_rename_head(_fetch(settings["url"]), model))
    _apply_settings(model, settings)
    return model


def inflate_pretrained(model, num_classes, settings):
    """reference torchvision_models.py:170-191: 2-D filters repeated along T (no 1/T scaling)."""
    assert num_classes == settings["num_classes"], \
        "num_classes should be {}, but is {}".format(settings["num_classes"], num_classes)
    target = model.state_dict()
    sd = _rename_head(_fetch(settings["url"]), model)
    for k, v in list(sd.items()):
        if k in target and v.shape != target[k].shape:
            sd[k] = v.unsqueeze(2).expand_as(target[k]).contiguous()
    model.load_state_dict(sd)
    _apply_settings(model, settings)
    return model


def _build(arch_name, num_classes, shortcut_type=None):
    if shortcut_type is not None and shortcut_type != ARCHS[arch_name].shortcut:
        key = "%s@%s" % (arch_name, shortcut_type)
        if key not in ARCHS:
            ARCHS[key] = dataclasses.replace(ARCHS[arch_name], shortcut=shortcut_type)
        arch_name = key
    return VideoResNet(arch_name, num_classes)


def _resnet3d_factory(name, default_shortcut, default_classes=400, default_pretrained="kinetics-400"):
    def factory(num_classes=default_classes, pretrained=default_pretrained, shortcut_type=default_shortcut, **kwargs):
        if kwargs:
            raise TypeError("%s() got unexpected keyword arguments %s" % (name, sorted(kwargs)))
        model = _build(name, num_classes, shortcut_type)
        if pretrained is not None:
            load_pretrained(model, num_classes, pretrained_settings[name][pretrained])
        return model
    factory.__name__ = name
    factory.__doc__ = "Constructs a %s model (reference pretorched/models/resnet3D.py)." % name
    return factory


def resnet3d10(num_classes=339, shortcut_type="B"):
    """reference resnet3D.py:242-246 (no pretrained weights exist)."""
    return _build("resnet3d10", num_classes, shortcut_type)


resnet3d18 = _resnet3d_factory("resnet3d18", "A")
resnet3d34 = _resnet3d_factory("resnet3d34", "A")
resnet3d50 = _resnet3d_factory("resnet3d50", "B")
resnet3d101 = _resnet3d_factory("resnet3d101", "B")
resnet3d152 = _resnet3d_factory("resnet3d152", "B")


def resnet3d200(num_classes=400, pretrained="kinetics-400", **kwargs):
    """reference resnet3D.py:301-308: `num_classes` is not forwarded to the constructor there
    (the network is built with the class default, 339) -- mirrored."""
    model = _build("resnet3d200", kwargs.pop("num_classes_override", 339), kwargs.pop("shortcut_type", "B"))
    if pretrained is not None:
        load_pretrained(model, num_classes, pretrained_settings["resnet3d200"][pretrained])
    return model


def resneti3d50(num_classes=400, pretrained="moments", shortcut_type="B"):
    """ResNet3D-50 initialised by inflating the 2-D Moments ResNet-50 (resnet3D.py:311-318)."""
    model = _build("resneti3d50", num_classes, shortcut_type)
    if pretrained is not None:
        inflate_pretrained(model, num_classes, pretrained_settings["resnet50"][pretrained])
    return model


def nonlocalresnet3d50(num_classes=339, num_nonlocal_blocks=5, pretrained="kinetics-400", shortcut_type="A"):
    """reference nonlocalnet.py:553-570.  As there: `num_classes` is accepted but NOT forwarded
    (always 339 logits); shortcut type 'A'; the checkpoint is loaded non-strictly."""
    if num_nonlocal_blocks == 5:
        nl = (0, 2, 3, 0)
    elif num_nonlocal_blocks == 10:
        nl = (0, 4, 6, 0)
    else:
        raise ValueError("num_nonlocal_blocks must be 5 or 10 (the reference fails with UnboundLocalError)")
    key = "nonlocalresnet3d50/%d" % num_nonlocal_blocks
    if key not in ARCHS:
        ARCHS[key] = dataclasses.replace(ARCHS["nonlocalresnet3d50"], nonlocal_layers=nl)
    model = _build(key, 339, shortcut_type)
    if pretrained is not None:
        settings = pretrained_settings["nonlocalresnet3d50"][pretrained]
        model.load_state_dict(_fetch(settings["url"]), strict=False)
        _apply_settings(model, settings)
    return model


def _nonlocal_factory(name, block, layers):
    def factory(**kwargs):
        """reference nonlocalnet.py:524-577: `NonLocalResNet3D(block, layers, **kwargs)` -- `nonlocal_layers` is a REQUIRED
        constructor argument these factories do not supply, so a bare call raises TypeError upstream and
        here; with `nonlocal_layers=[...]` (and optionally shortcut_type='A', num_classes=339) it builds, as upstream."""
        if "nonlocal_layers" not in kwargs:
            raise TypeError("NonLocalResNet3D.__init__() missing 1 required positional argument: 'nonlocal_layers'")
        nl = tuple(int(v) for v in kwargs.pop("nonlocal_layers"))
        shortcut_type = kwargs.pop("shortcut_type", "A")
        num_classes = kwargs.pop("num_classes", 339)
        if kwargs:
            raise TypeError("NonLocalResNet3D.__init__() got an unexpected keyword argument %r" % sorted(kwargs)[0])
        if len(nl) != 4:
            raise IndexError("list index out of range")       # what `nonlocal_layers[3]` raises upstream (nonlocalnet.py:443)
        key = "%s/nl%s" % (name, "-".join(str(v) for v in nl))
        if key not in ARCHS:
            ARCHS[key] = Arch(block, layers, "A", nonlocal_layers=nl)
        return _build(key, num_classes, shortcut_type)
    factory.__name__ = name
    return factory


nonlocalresnet3d18 = _nonlocal_factory("nonlocalresnet3d18", "basic", (2, 2, 2, 2))
nonlocalresnet3d34 = _nonlocal_factory("nonlocalresnet3d34", "basic", (3, 4, 6, 3))
nonlocalresnet3d101 = _nonlocal_factory("nonlocalresnet3d101", "bottleneck", (3, 4, 23, 3))


def _r2plus1d_factory(name):
    def factory(num_classes=339, shortcut_type="B"):
        return _build(name, num_classes, shortcut_type)
    factory.__name__ = name
    factory.__doc__ = "Constructs a %s model (reference pretorched/models/r2plus1d.py:113-152)." % name
    return factory


r2plus1d10 = _r2plus1d_factory("r2plus1d10")
r2plus1d18 = _r2plus1d_factory("r2plus1d18")
r2plus1d34 = _r2plus1d_factory("r2plus1d34")
r2plus1d50 = _r2plus1d_factory("r2plus1d50")


def _resnext3d_factory(name):
    def factory(**kwargs):
        """reference resnext3D.py:213-252: `ResNeXt3D(ResNeXtBottleneck, layers, **kwargs)` with
        shortcut_type='B', cardinality=32, num_classes=400 defaults; no pretrained argument upstream."""
        shortcut_type = kwargs.pop("shortcut_type", "B")
        cardinality = kwargs.pop("cardinality", 32)
        num_classes = kwargs.pop("num_classes", 400)
        if kwargs:
            raise TypeError("%s() got unexpected keyword arguments %s" % (name, sorted(kwargs)))
        key = name if cardinality == 32 else "%s/c%d" % (name, cardinality)
        if key not in ARCHS:
            ARCHS[key] = dataclasses.replace(ARCHS[name], cardinality=cardinality)
        return _build(key, num_classes, shortcut_type)
    factory.__name__ = name
    return factory


resnext3d10 = _resnext3d_factory("resnext3d10")
resnext3d18 = _resnext3d_factory("resnext3d18")
resnext3d34 = _resnext3d_factory("resnext3d34")
resnext3d50 = _resnext3d_factory("resnext3d50")
resnext3d101 = _resnext3d_factory("resnext3d101")
resnext3d152 = _resnext3d_factory("resnext3d152")
resnext3d200 = _resnext3d_factory("resnext3d200")


def _preact_factory(name):
    def factory(num_classes=339, shortcut_type="B"):
        return _build(name, num_classes, shortcut_type)
    factory.__name__ = name
    factory.__doc__ = ("Constructs a %s model (reference pretorched/models/pre_act_resnet3D.py:103-142: "
                       "PreActivationResNet3D(block, layers, **kwargs) over ResNet3D's constructor)." % name)
    return factory


preact_resnet3d10 = _preact_factory("preact_resnet3d10")
preact_resnet3d18 = _preact_factory("preact_resnet3d18")
preact_resnet3d34 = _preact_factory("preact_resnet3d34")
preact_resnet3d50 = _preact_factory("preact_resnet3d50")
preact_resnet3d101 = _preact_factory("preact_resnet3d101")
preact_resnet3d152 = _preact_factory("preact_resnet3d152")
preact_resnet3d200 = _preact_factory("preact_resnet3d200")


def _mvresnet_factory(name):
    def factory(num_classes=339, shortcut_type="B"):
        return _build(name, num_classes, shortcut_type)
    factory.__name__ = name
    factory.__doc__ = ("Constructs a %s model (reference pretorched/models/multiview.py:95-140: MVResNet(block, layers, "
                       "**kwargs) -- ResNet3D with every convolution a MultiViewConv; module-level upstream)." % name)
    return factory


mvresnet10 = _mvresnet_factory("mvresnet10")
mvresnet18 = _mvresnet_factory("mvresnet18")
mvresnet34 = _mvresnet_factory("mvresnet34")
mvresnet50 = _mvresnet_factory("mvresnet50")
mvresnet101 = _mvresnet_factory("mvresnet101")
mvresnet152 = _mvresnet_factory("mvresnet152")
mvresnet200 = _mvresnet_factory("mvresnet200")


def wideresnet3d50(num_classes=400, pretrained="kinetics-400", shortcut_type="B", k=2):
    """reference wideresnet3D.py:202-210 (module-level upstream; no checkpoint URL is published for it)."""
    key = "wideresnet3d50" if k == 2 else "wideresnet3d50/k%d" % k
    if key not in ARCHS:
        ARCHS[key] = dataclasses.replace(ARCHS["wideresnet3d50"], k=k)
    model = _build(key, num_classes, shortcut_type)
    if pretrained is not None:
        settings = pretrained_settings.get("wideresnet3d50", {}).get(pretrained)
        if settings is None:
            raise PtxError("no pretrained weights are published for wideresnet3d50 (the reference's settings table "
                           "has no entry either)")
        load_pretrained(model, num_classes, settings)
    return model


def nonlocal_r2plus1d50(num_classes=339):
    """BASELINE.json config 3 ("resnet2p1d50 + NLBlock"): no reference model combines (2+1)D convs
    with NL blocks; this is the composition validated in SURVEY.md row A9 (shortcut 'B')."""
    return _build("nonlocal_r2plus1d50", num_classes)


def _resnet2d_factory(name):
    def factory(num_classes=1000, pretrained="imagenet"):
        model = _build(name, num_classes)
        if pretrained is not None:
            load_pretrained(model, num_classes, pretrained_settings[name][pretrained])
        return model
    factory.__name__ = name
    factory.__doc__ = ("2-D %s (reference torchvision_models.py:484-536, arithmetic in torchvision there), "
                       "run as the T == 1 case of the same HIP engine." % name)
    return factory


resnet18 = _resnet2d_factory("resnet18")
resnet34 = _resnet2d_factory("resnet34")
resnet50 = _resnet2d_factory("resnet50")
resnet101 = _resnet2d_factory("resnet101")
resnet152 = _resnet2d_factory("resnet152")


def trn(num_classes=339, num_segments=8, consensus="MSTRN", arch="resnet50", pretrained="moments",
        frame_bottleneck_dim=1024, video_feature_dim=1024):
    """reference trn.py:345-355.  As there, `consensus`, the two widths and `pretrained=None` are
    NOT forwarded to TRN (which therefore uses its own defaults: 'HTRN', 1024/1024, a 'moments'
    backbone download); construct `TRN(...)` directly to choose them."""
    if pretrained:
        settings = pretrained_settings["trn"][pretrained]
        assert num_classes == settings["num_classes"], \
            "num_classes should be {}, but is {}".format(settings["num_classes"], num_classes)
        model = TRN(num_classes=num_classes, num_segments=num_segments, arch=arch)
        model.load_state_dict(_fetch(settings["url"] or None))
    else:
        model = TRN(num_classes=num_classes, num_segments=num_segments, arch=arch)
    return model


model_names = ["resnet3d10", "resnet3d18", "resnet3d34", "resnet3d50", "resnet3d101", "resnet3d152",
               "resnet3d200", "resneti3d50", "nonlocalresnet3d50", "nonlocalresnet3d18", "nonlocalresnet3d34",
               "nonlocalresnet3d101", "r2plus1d10", "r2plus1d18",
               "r2plus1d34", "r2plus1d50", "nonlocal_r2plus1d50", "resnet18", "resnet34", "resnet50",
               "resnet101", "resnet152", "trn", "i3d", "resnext3d10", "resnext3d18", "resnext3d34", "resnext3d50",
               "resnext3d101", "resnext3d152", "resnext3d200", "wideresnet3d50", "preact_resnet3d10", "preact_resnet3d18", "preact_resnet3d34",
               "preact_resnet3d50", "preact_resnet3d101", "preact_resnet3d152", "preact_resnet3d200",
               "mvresnet10", "mvresnet18", "mvresnet34", "mvresnet50", "mvresnet101", "mvresnet152", "mvresnet200"]
