"""Round 4: where does the split-vs-full disagreement of VERDICT r3 (weak #1) come from?

One model (default synthetic recipe unless RECIPE=cfg3), one seeded batch.  Ground truth = the oracle in fp64 on two clips
(one of each half); next to it the oracle in fp32 (the network's own fp32 noise floor on the CPU) and the HIP engine as
B=8 (one plan), B=4+4 (one plan, sequential), B=1x8, two instances on two streams.  With BLOCKS=1 every residual block's
output of the B=8 and the B=4 plan is compared with the fp64 oracle's, so a wrong tile / chain / attention variant shows
up at the block where it enters.

    python scripts/gpu_split_vs_full.py nonlocal_r2plus1d50 8x3x32x112x112
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd import engine as E  # noqa: E402
from pretorched_x_amd.testing import synth_clips, synth_state_dict  # noqa: E402
from oracle import functional as OF  # noqa: E402  (the checker, never the thing measured)

DEV = os.environ.get("DEV", "cuda:0")
arch = sys.argv[1] if len(sys.argv) > 1 else "nonlocal_r2plus1d50"
shape = tuple(int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "8x3x32x112x112").split("x"))
kw = dict(num_classes=339, pretrained=None) if arch in ("resnet3d50",) else dict(pretrained=None) if arch == "nonlocalresnet3d50" else dict(num_classes=339)
recipe = {"cfg3": dict(inner_bn_damp=0.9, nl_bn_damp=0.05),
          # the full-strength-NL fixture recipe (tests/golden/make_golden.py RECIPES): soft attention, W.1 undamped
          "fullnl": dict(inner_bn_damp=0.85, last_bn_damp=0.55, nl_bn_damp=1.0, nl_embed_damp={"layer2": 0.3, "layer3": 0.1}),
          }.get(os.environ.get("RECIPE", ""), {})
CLIPS = [0, shape[0] // 2 + 1]
BLOCKS = os.environ.get("BLOCKS", "0") == "1"

# ---- record every conv output of a plan by label (debug only: the product keeps raw pointers)
acts_of = {}
_conv, _chain = E.Plan.conv, E.Plan.conv_chain


def conv(self, x, pk, *a, **k):
    r = _conv(self, x, pk, *a, **k)
    acts_of.setdefault(id(self), {})[k.get("label", "conv")] = r[0] if isinstance(r, tuple) else r
    return r


def conv_chain(self, *a, **k):
    r = _chain(self, *a, **k)
    if r is not None:
        acts_of.setdefault(id(self), {})[k.get("label", "chain")] = r[0]
    return r


E.Plan.conv, E.Plan.conv_chain = conv, conv_chain


def build():
    m = ptx.__dict__[arch](**kw)
    sd = synth_state_dict(m.state_dict(), 1234, **recipe)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


model, sd = build()
x = synth_clips(shape[0], shape[2], shape[3], 99)
xd = x.to(DEV)
cfg = OF.ARCHS[arch]

# ---- CPU: fp64 ground truth + fp32 oracle on the probe clips, block outputs recorded
blocks64, blocks32 = {}, {}
_blk = OF._block


def run_oracle(dtype, store):
    def rec(cfg_, sd_, x_, p, *a):
        out = _blk(cfg_, sd_, x_, p, *a)
        store[p] = out
        return out
    OF._block = rec
    try:
        sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        with torch.no_grad():
            return OF.forward(cfg, sdd, x[CLIPS].to(dtype))
    finally:
        OF._block = _blk


t0 = time.time()
ref64 = run_oracle(torch.float64, blocks64)
ref32 = run_oracle(torch.float32, blocks32)
print("oracle fp64 + fp32 on clips %s: %.1f s; max|logit| %.2f; fp32 oracle vs fp64: %.3e" % (
    CLIPS, time.time() - t0, ref64.abs().max().item(), (ref32.double() - ref64).abs().max().item()), flush=True)


def err(name, out):
    o = out.detach().double().cpu()
    e = (o[CLIPS] - ref64).abs().max().item()
    am = bool(torch.equal(o[CLIPS].argmax(1), ref64.argmax(1)))
    print("  %-34s max|d| vs fp64 oracle %.3e  (rel %.2e)  argmax_equal=%s" % (name, e, e / ref64.abs().max().item(), am), flush=True)
    return o


with torch.no_grad():
    full = err("B=8 one plan", model(xd))
    h = shape[0] // 2
    seq4 = err("B=%d + B=%d sequential, one model" % (h, h), torch.cat([model(xd[:h].contiguous()), model(xd[h:].contiguous())], 0))
    seq1 = err("B=1 x %d" % shape[0], torch.cat([model(xd[i:i + 1].contiguous()) for i in range(shape[0])], 0))
    if DEV == "cpu":
        sys.exit(0)
    halves = [build()[0] for _ in range(2)]
    xs = [xd[:h], xd[h:]]
    for m_, xi in zip(halves, xs):
        m_(xi)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(2)]
    cur = torch.cuda.current_stream()
    outs = []
    for m_, xi, st in zip(halves, xs, streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(m_(xi))
    for st in streams:
        cur.wait_stream(st)
    torch.cuda.synchronize()
    two = err("B=%d + B=%d two instances, two streams" % (h, h), torch.cat(outs, 0))
    again = model(xd).double().cpu()
print("pairwise over ALL clips: |seq4-full| %.3e  |seq1-full| %.3e  |two-full| %.3e  |two-seq4| %.3e  |full-full again| %.3e" % (
    (seq4 - full).abs().max().item(), (seq1 - full).abs().max().item(), (two - full).abs().max().item(),
    (two - seq4).abs().max().item(), (again - full).abs().max().item()), flush=True)

if BLOCKS:
    eng = model.engine()
    plans = {k[0][0]: p for k, p in eng._plans.items()}
    with torch.no_grad():
        model(xd)
        model(xd[:h].contiguous())      # leaves clips 0..h-1 in the B=h plan's buffers
        torch.cuda.synchronize()
    print("%-28s %12s %12s %12s %12s" % ("block output (clip 0)", "max|ref|", "fp32 oracle", "HIP B=%d" % shape[0], "HIP B=%d" % h))
    for p in blocks64:
        want = blocks64[p][0]                                   # clip 0: [C,T,H,W]
        row = [want.abs().max().item(), (blocks32[p][0].double() - want).abs().max().item()]
        for b in (shape[0], h):
            pl = plans.get(b)
            labs = acts_of.get(id(pl), {})
            live = set(getattr(s, "label", "") for s in pl.all_convs())
            cand = [l for l in labs if l.startswith(p + ".") and l in live]
            if not cand:
                row.append(float("nan"))
                continue
            a = labs[cand[-1]]
            got = a.t[0, ..., :a.C].permute(3, 0, 1, 2).double().cpu()
            row.append((got - want).abs().max().item())
        print("%-28s %12.4e %12.3e %12.3e %12.3e" % (p, *row), flush=True)
