"""Per-launch rows of one zoo case (scripts/gpu_zoo_bench.py CASES) at the engine's own tile choices.
usage: [PTX_PRECISION=x3] python scripts/gpu_case_rows.py "<case substring>" [n_slowest]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
args = sys.argv[1:]
sys.argv = sys.argv[:1]
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import synth_state_dict  # noqa: E402
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_zoo_bench.py")).read().split("only = sys.argv[1:]")[0]
ns = {"__file__": os.path.abspath(__file__)}
exec(compile(src, "zoo_cases", "exec"), ns)
name, build, recipe, shape = ([c for c in ns["CASES"] if args[0] == c[0]] or [c for c in ns["CASES"] if args[0] in c[0]])[0]
m = build()
m.load_state_dict(synth_state_dict(m.state_dict(), 1234, **recipe))
m = m.cuda().eval()
if os.environ.get("PTX_PRECISION"):
    m.engine().precision = os.environ["PTX_PRECISION"]
x = torch.randn(*shape, device="cuda")
eng = m.engine()
for _ in range(3):
    m(x)
torch.cuda.synchronize()
plan = list(eng._plans.values())[-1]
rows = eng.profile_steps(plan, iters=5)
print("%s %s precision=%s: %d launches, %.3f ms summed" % (name, "x".join(map(str, shape)), eng.precision, len(rows), sum(r[4] for r in rows)))
for r in sorted(rows, key=lambda r: -r[4])[:int(args[1]) if len(args) > 1 else 20]:
    print("%-34s %-6s %9.4f ms  %s" % (r[0][:34], r[1], r[4], r[5]))
