"""Throughput of every model family behind the engine beyond the headline config (bench.py covers
resnet3d50 / config 2): config 3 and its parents, SlowFast, I3D (config 4 share of one GPU), TRN, 2-D.
Algorithmic FLOPs = 2 x the MACs of the compiled plan's conv launches (padding taps counted) + heads
ignored; wall-clock with the queue drained on both sides.  Writes gpurun_out/zoo_bench.json."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import BIGGAN_RECIPE, I3D_RECIPE, synth_state_dict  # noqa: E402

CASES = [
    # name, builder, recipe, input shape, clips per step
    ("nonlocal_r2plus1d50 (cfg3)", lambda: ptx.nonlocal_r2plus1d50(339), dict(inner_bn_damp=0.9, nl_bn_damp=0.05), (8, 3, 32, 112, 112)),
    ("r2plus1d50", lambda: ptx.r2plus1d50(400), dict(inner_bn_damp=0.9), (8, 3, 32, 112, 112)),
    ("nonlocalresnet3d50", lambda: ptx.nonlocalresnet3d50(pretrained=None), dict(last_bn_damp=0.65, nl_bn_damp=0.05), (8, 3, 32, 112, 112)),
    ("slowfast resnet50 SF", lambda: ptx.slowfast.resnet50(num_classes=400), {}, (8, 3, 64, 224, 224)),
    ("slowfast resnet50 S", lambda: ptx.slowfast.resnet50(mode="S", num_classes=400), {}, (8, 3, 64, 224, 224)),
    ("slowfast resnet50 F", lambda: ptx.slowfast.resnet50(mode="F", num_classes=400), {}, (8, 3, 64, 224, 224)),
    ("i3d (cfg4 per-GPU share)", lambda: ptx.i3d(400), I3D_RECIPE, (2, 3, 64, 224, 224)),
    ("i3d batch 8", lambda: ptx.i3d(400), I3D_RECIPE, (8, 3, 64, 224, 224)),
    ("resnet3d101", lambda: ptx.resnet3d101(num_classes=400, pretrained=None), {}, (8, 3, 16, 224, 224)),
    ("resnext3d101", lambda: ptx.resnext3d101(num_classes=400), dict(last_bn_damp=2.0), (8, 3, 16, 224, 224)),
    ("wideresnet3d50", lambda: ptx.wideresnet3d50(num_classes=400, pretrained=None), {}, (8, 3, 16, 224, 224)),
    ("preact_resnet3d50", lambda: ptx.preact_resnet3d50(num_classes=339), {}, (8, 3, 16, 224, 224)),
    ("resnet3d18", lambda: ptx.resnet3d18(num_classes=400, pretrained=None), {}, (8, 3, 16, 224, 224)),
    ("resnet50 2-D", lambda: ptx.resnet50(num_classes=1000, pretrained=None), dict(last_bn_damp=0.7), (64, 3, 224, 224)),
    ("biggan-deep-256 G (cfg5, fp32)", lambda: ptx.biggan_deep(256), BIGGAN_RECIPE, (64, 128)),
    ("biggan-deep-256 G fp16 operands", lambda: ptx.biggan_deep(256, precision="fp16"), BIGGAN_RECIPE, (64, 128)),
    ("TRN resnet50 x8 frames", lambda: ptx.TRN(339, num_segments=8, consensus="MSTRN", pretrained=None), dict(last_bn_damp=0.7), (8, 8, 3, 224, 224)),
]
# CPU leg (the reference path's restatement, oracle/): bounded sample of the same workload, like bench.py
from oracle import functional as OF  # noqa: E402
from oracle import biggan_standin as BG, i3d_standin as I3  # noqa: E402


def _cpu_fn(name, sd):
    if "cfg3" in name:
        return lambda x: OF.forward(OF.ARCHS["nonlocal_r2plus1d50"], sd, x)
    for arch in ("r2plus1d50", "nonlocalresnet3d50", "resnet3d101", "resnet3d18", "resnext3d101", "wideresnet3d50",
                 "preact_resnet3d50"):
        if name.startswith(arch):
            return lambda x, a=arch: OF.forward(OF.ARCHS[a], sd, x)
    if name.startswith("slowfast"):
        mode = {"SF": "sf", "S": "s", "F": "f"}[name.split()[-1]]
        return lambda x: OF.slowfast_forward(sd, x, "bottleneck", [3, 4, 6, 3], mode)
    if name.startswith("i3d"):
        return lambda x: I3.forward(sd, x)
    if name.startswith("resnet50 2-D"):
        return lambda x: OF.forward(OF.ARCHS["resnet50"], sd, x)
    if name.startswith("TRN"):
        return lambda x: OF.trn_forward(OF.ARCHS["resnet50"], sd, x, 8, "MSTRN")
    if name.startswith("biggan"):
        return lambda z: BG.forward(sd, z, sd["shared.weight"][:z.shape[0]])
    return None


CPU_THREADS = int(os.environ.get("ZOO_CPU_THREADS", "32"))
only = sys.argv[1:]
rows = []
for name, build, recipe, shape in CASES:
    if only and not any(o in name for o in only):
        continue
    torch.manual_seed(0)
    m = build()
    sd_cpu = synth_state_dict(m.state_dict(), 1234, **recipe)
    m.load_state_dict(sd_cpu)
    cpu_units = None
    fn = _cpu_fn(name, sd_cpu)
    if fn is not None and not os.environ.get("ZOO_NO_CPU"):
        torch.set_num_threads(CPU_THREADS)
        nb = min(2, shape[0])
        xc = torch.randn(nb, *shape[1:])
        with torch.no_grad():
            fn(xc[:1])                                   # warm-up (oneDNN primitive creation)
            t0 = time.perf_counter()
            fn(xc)
            cpu_units = nb / (time.perf_counter() - t0)
    m = m.cuda().eval()
    x = torch.randn(*shape, device="cuda")
    if name.startswith("biggan"):             # generator: (z, shared(labels)) -> images
        yemb = m.shared(torch.randint(0, 1000, (shape[0],), device="cuda"))
        gen, m_call = m, None
        m = type("G", (), {"__call__": lambda self, z: gen(z, yemb), "engine": gen.engine})()
    t0 = time.time()
    y = m(x)                                  # compiles the plan and times untuned tile choices
    torch.cuda.synchronize()
    t_first = time.time() - t0
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        y = m(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    eng = (m.base_model if hasattr(m, "base_model") else m).engine()
    plans = list(eng._plans.values())
    from pretorched_x_amd.engine import StemStep, StemF32Step       # the direct stems are convs too (not in conv_steps)
    gflop = sum(2e-9 * s.macs for p in plans for s in p.all_convs())     # the launches that run (chained pairs count once)
    rows.append(dict(model=name, input=list(shape), ms_per_step=round(ms, 3), units_per_s=round(shape[0] * 1e3 / ms, 1),
                     gflop_per_step=round(gflop, 1), tflops=round(gflop / ms, 1), frac_fp32_mfma=round(gflop / ms / 157.3, 3),
                     conv_launches=sum(len(p.all_convs()) for p in plans), first_call_s=round(t_first, 1),
                     cpu_units_per_s=None if cpu_units is None else round(cpu_units, 2), cpu_threads=CPU_THREADS,
                     finite=bool(torch.isfinite(y).all())))
    print("%-28s %-22s %9.3f ms  %8.1f /s  %8.1f GFLOP  %6.1f TF (%4.1f%%)  launches %d  cpu %s /s on %d threads" % (
        name, "x".join(map(str, shape)), ms, shape[0] * 1e3 / ms, gflop, gflop / ms, gflop / ms / 1.573,
        rows[-1]["conv_launches"], "-" if cpu_units is None else "%.2f" % cpu_units, CPU_THREADS), flush=True)
    del m, x, y
    torch.cuda.empty_cache()
if os.environ.get("PTX_TUNED_OUT"):               # every tile choice made in this process (all models) -> one table
    from pretorched_x_amd.engine import save_tuned_table
    save_tuned_table(os.environ["PTX_TUNED_OUT"])
json.dump(dict(peak_tflops=157.3, note="clips (videos / images for TRN / 2-D) per second on one MI355X, fp32 MFMA; "
               "FLOPs = 2 x MACs of the plan's conv launches; cpu_units_per_s = oracle/ (CPU restatement of the "
               "reference path) on a 2-unit sample of the same workload", rows=rows), open(os.path.join(ROOT, "gpurun_out", "zoo_bench.json"), "w"), indent=1)
