"""accelerate(model): put the HIP engine behind a model instance the REFERENCE built.

The reference's users hold instances of its own classes -- `pretorched.__dict__['resnet3d50'](...)` returns a
`pretorched.models.resnet3D.ResNet3D` whose `features / logits / forward` were patched onto the CLASS by `modify_resnets`
(/root/reference/pretorched/models/torchvision_models.py:443-481), `NonLocalResNet3D` carries its own
(nonlocalnet.py:487-508), `R2Plus1D` keeps `ResNet3D.forward` and `fc` (r2plus1d.py:99-110).  `accelerate` walks such an
instance, derives the Arch (block class, blocks per stage, shortcut type, conv kind, non-local placement) from the module
tree itself, builds a plan-owning twin whose children ARE the instance's submodules -- no parameter is copied: `.cuda()`,
`load_state_dict`, optimiser steps and `last_linear` replacement on the instance are all seen by the engine -- and binds
`features / logits / forward` on the INSTANCE (never on the class: other instances of the reference class are untouched).
Calls outside the engine's contract (train() mode, autograd, CPU tensors) run the reference's own methods, unchanged.
"""
import torch
import torch.nn as nn

from . import eager
from ._lib import PtxError
from .zoo import Arch, Bag, VideoResNet

_BLOCK_ATTRS = ("conv1", "bn1", "conv2", "bn2", "conv3", "bn3")


def _is_factored(conv):
    return hasattr(conv, "spatial_conv") and hasattr(conv, "temporal_conv")


def _arch_of(model):
    """Arch of a reference ResNet3D / R2Plus1D / NonLocalResNet3D instance, read off its module tree."""
    missing = [n for n in ("conv1", "bn1", "layer1", "layer2", "layer3", "layer4") if not hasattr(model, n)]
    if missing:
        raise PtxError("accelerate: %s is not a ResNet3D-family model (no %s)" % (type(model).__name__, ", ".join(missing)))
    layers = tuple(len(getattr(model, "layer%d" % i)) for i in range(1, 5))
    first = model.layer1[0]
    names = {type(b).__name__ for i in range(1, 5) for b in getattr(model, "layer%d" % i)}
    if any("PreAct" in n or "ResNeXt" in n or "Wide" in n for n in names):
        raise PtxError("accelerate covers ResNet3D / R2Plus1D / NonLocalResNet3D instances; build %s through pretorched_x_amd" % sorted(names))
    if hasattr(first, "conv3") and hasattr(first, "bn3"):
        block = "bottleneck"
    elif hasattr(first, "conv2") and hasattr(first, "bn2") and not hasattr(first, "conv3"):
        block = "basic"
    else:
        raise PtxError("accelerate: unrecognised residual block %s" % type(first).__name__)
    shortcut = "B"
    for i in range(1, 5):
        for b in getattr(model, "layer%d" % i):
            ds = getattr(b, "downsample", None)
            if ds is not None:
                shortcut = "B" if isinstance(ds, nn.Module) else "A"       # nn.Sequential(conv, bn) | partial(downsample_basic_block)
                break
        else:
            continue
        break
    conv = "2p1d" if _is_factored(model.conv1) else "3d"
    if getattr(model, "last_linear", None) is not None:
        head = "last_linear"
    elif getattr(model, "fc", None) is not None:
        head = "fc"
    else:
        raise PtxError("accelerate: the model has neither `last_linear` nor `fc`")
    dims = 2 if isinstance(model.conv1, nn.Conv2d) else 3
    if dims == 2:
        raise PtxError("accelerate covers the 3-D families; the 2-D ResNets are torchvision modules upstream")
    return Arch(block, layers, shortcut, conv=conv, head=head, dims=dims)


def _nl_view(nl):
    """The engine's view of a reference _NonLocalBlockND (nonlocalnet.py:51-131): the same child modules plus the
    constructor flags the reference does not keep as attributes."""
    v = Bag()
    v.g, v.W, v.theta, v.phi = nl.g, nl.W, nl.theta, nl.phi
    if getattr(nl, "concat_project", None) is not None:
        v.concat_project = nl.concat_project
    v.mode = nl.mode
    v.sub_sample = bool(nl.sub_sample)
    v.dimension = int(getattr(nl, "dimension", 3))
    v.bn_layer = isinstance(nl.W, nn.Sequential)
    return v


class AdoptedResNet(VideoResNet):
    """VideoResNet over ANOTHER model's modules (see accelerate).  Never constructed from an Arch name."""

    def __init__(self, source, arch, engine=None):
        nn.Module.__init__(self)
        object.__setattr__(self, "_source", source)           # not a child: the source owns this twin, not vice versa
        self.arch_name = "adopted:%s" % type(source).__name__
        self.arch = arch
        self.conv1, self.bn1 = source.conv1, source.bn1
        self.relu, self.maxpool, self.avgpool = source.relu, source.maxpool, source.avgpool
        cin = None
        for li in range(1, 5):
            blocks = []
            for b in getattr(source, "layer%d" % li):
                blk = Bag()
                for n in _BLOCK_ATTRS:
                    if hasattr(b, n):
                        setattr(blk, n, getattr(b, n))
                ds = getattr(b, "downsample", None)
                blk.downsample = ds if isinstance(ds, nn.Module) else None
                blk.has_shortcut = ds is not None
                blk.stride = b.stride
                blk.has_nl = hasattr(b, "nonlocalblock")
                if blk.has_nl:
                    blk.nonlocalblock = _nl_view(b.nonlocalblock)
                blocks.append(blk)
            setattr(self, "layer%d" % li, nn.ModuleList(blocks))
        del cin
        self.fc = None
        self.train(False)
        if engine is None:
            self._init_engine()
        else:
            self._engine = engine             # a DataParallel replica's twin shares the original's Engine

    # the classifier is read from the SOURCE at call time: users replace `last_linear` on the instance they hold
    @property
    def head_module(self):
        return getattr(self._source, self.arch.head)

    def train(self, mode=True):
        # the twin's children are the source's modules: their mode is the source's business
        self.training = False
        return self


_METHODS = ("features", "logits", "forward")


def _twin_of(model):
    """The engine-owning twin OF THIS INSTANCE.  A copy of an accelerated model (copy.deepcopy, an unpickled torch.save,
    a DataParallel replica) carries a copy of the original's __dict__ -- its twin must be one over ITS OWN modules, never
    the original's (silent wrong-model / cross-device execution), so it is rebuilt on first use."""
    twin = model.__dict__.get("_ptx_twin")
    if twin is None or twin.__dict__.get("_source") is not model:
        twin = AdoptedResNet(model, _arch_of(model))
        object.__setattr__(model, "_ptx_twin", twin)
        object.__setattr__(model, "_engine", twin._engine)
    return twin


class _Bound:
    """`model.<name>` of an accelerated instance: a picklable, deep-copyable callable stored in the instance __dict__ that
    resolves the twin from the instance it is bound to AT CALL TIME (a closure over the original model and twin, as in
    round 5, kept serving the original after a copy)."""
    __slots__ = ("model", "name")

    def __init__(self, model, name):
        self.model, self.name = model, name

    def __deepcopy__(self, memo):
        import copy
        return _Bound(copy.deepcopy(self.model, memo), self.name)       # memo holds the copy under construction

    def __reduce__(self):
        return (_Bound, (self.model, self.name))

    @property
    def __wrapped__(self):
        return getattr(type(self.model), self.name)

    def __call__(self, *a, **k):
        model, name = self.model, self.name
        if name == "engine":
            return _twin_of(model)._engine
        if name == "_replicate_for_data_parallel":
            # nn.DataParallel (examples/imagenet_eval.py:136) copies __dict__ into every replica: re-bind the replica's
            # methods to the replica, and give it a twin over ITS modules (the per-device broadcast copies) that shares the
            # original's Engine -- plans are keyed by (shape, device), weight identity stays the original's (Engine.owner)
            rep = type(model)._replicate_for_data_parallel(model)
            twin = AdoptedResNet(rep, _twin_of(model).arch, engine=_twin_of(model)._engine)
            object.__setattr__(twin, "_is_replica", True)
            object.__setattr__(rep, "_ptx_twin", twin)
            _bind(rep)
            return rep
        ref_fn = getattr(type(model), name)
        x = a[0] if a else None
        if len(a) != 1 or k or eager.wanted(model, x):
            return ref_fn(model, *a, **k)                    # the reference's own code: train() / autograd / CPU
        twin = _twin_of(model)
        return getattr(twin._engine, name)(twin, x)


def _bind(model):
    cls = type(model)
    for name in _METHODS:
        if callable(getattr(cls, name, None)):
            object.__setattr__(model, name, _Bound(model, name))
    object.__setattr__(model, "engine", _Bound(model, "engine"))
    object.__setattr__(model, "_replicate_for_data_parallel", _Bound(model, "_replicate_for_data_parallel"))


def accelerate(model):
    """Bind the MI355X engine to `model`, an instance of the reference's ResNet3D / R2Plus1D / NonLocalResNet3D
    (/root/reference/pretorched/models/resnet3D.py:146, r2plus1d.py:99, nonlocalnet.py:423).  Returns `model`, with
    `features`, `logits` (when its class has them) and `forward` bound on the instance; `model.engine()` is the Engine.
    Copies stay correct: copy.deepcopy / torch.save + torch.load give an instance with its own twin and engine (built on
    first use), nn.DataParallel replicas run their own device's weights through the original's Engine."""
    if isinstance(model, VideoResNet):
        return model                                           # already engine-backed
    if not isinstance(model, nn.Module):
        raise PtxError("accelerate expects an nn.Module")
    twin = AdoptedResNet(model, _arch_of(model))
    object.__setattr__(model, "_ptx_twin", twin)
    object.__setattr__(model, "_engine", twin._engine)         # eager.wanted(model, x) reads the autograd opt-in here
    _bind(model)
    return model
