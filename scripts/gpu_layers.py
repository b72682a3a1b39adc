"""Per-layer view of one model: tuner choices (full log) and conv launch times.
usage: gpu_layers.py <case substring of scripts/gpu_zoo_bench.py CASES> [n_slowest]"""
import os
import sys
import collections
import torch
sys.path.insert(0, ".")
sys.argv, args = sys.argv[:1], sys.argv[1:]
os.environ["PTX_TUNE_LOG"] = "gpurun_out/tune_layers.log"
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import synth_state_dict  # noqa: E402
import importlib.util  # noqa: E402
spec = importlib.util.spec_from_file_location("zoo", "scripts/gpu_zoo_bench.py")
src = open("scripts/gpu_zoo_bench.py").read().split("only = sys.argv[1:]")[0]
ns = {"__file__": os.path.abspath("scripts/gpu_zoo_bench.py")}
exec(compile(src, "zoo_cases", "exec"), ns)
name, build, recipe, shape = [c for c in ns["CASES"] if args[0] in c[0]][0]
m = build()
m.load_state_dict(synth_state_dict(m.state_dict(), 1234, **recipe))
m = m.cuda().eval()
x = torch.randn(*shape, device="cuda")
eng = m.engine()
eng.auto_tune = False
m(x)
plan = list(eng._plans.values())[0]
eng.autotune(m, x, iters=3, plan=plan)
rows = collections.defaultdict(list)
for l in open(os.environ["PTX_TUNE_LOG"]):
    f = l.rstrip("\n").split("\t")
    rows[(f[0], f[1], f[2], f[3])].append((float(f[6].split()[0]), f[4], f[5]))
tot = 0.0
out = []
for stp in plan.conv_steps:
    pass
seen = set()
for k, v in rows.items():
    v.sort()
    best = v[0]
    mf = [r for r in v if "direct" not in r[1]]
    dr = [r for r in v if "direct" in r[1]]
    n = sum(1 for s in plan.conv_steps if s.label == k[0]) or 1
    out.append((best[0], k, best, mf[0] if mf else None, dr[0] if dr else None))
out.sort(key=lambda r: -r[0])
print("%s %s: %d conv launches" % (name, "x".join(map(str, shape)), len(plan.conv_steps)))
for t, k, best, mf, dr in out[:int(args[1]) if len(args) > 1 else 25]:
    print("%-30s %-10s %-6s %-8s best %.4f ms %-26s %-8s | mfma %s | direct %s" % (
        k[0][:30], k[1], k[2], k[3], best[0], best[1], best[2], ("%.4f %s" % (mf[0], mf[1])) if mf else "-",
        ("%.4f %s" % (dr[0], dr[1])) if dr else "-"))
