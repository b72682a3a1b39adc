"""Does running two half-batches on two HIP streams beat one full batch?  The small-M layers (layer3 / layer4: 392 tiles on
256 CUs, 10-25 us kernels) leave CUs idle; a second stream's kernels could fill them.  Two model instances (own plans and
buffers), 4 clips each, against one instance with 8 clips."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pretorched_x_amd as ptx  # noqa: E402
from pretorched_x_amd.testing import synth_clips, synth_state_dict  # noqa: E402

DEV = "cuda:0"
arch, shape = (sys.argv[1], tuple(int(v) for v in sys.argv[2].split("x"))) if len(sys.argv) > 2 else ("resnet3d50", (8, 3, 16, 224, 224))
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
kw = dict(num_classes=339, pretrained=None) if arch == "resnet3d50" else dict(num_classes=339)


def build():
    m = ptx.__dict__[arch](**kw)
    m.load_state_dict(synth_state_dict(m.state_dict(), 1234))
    return m.to(DEV).eval()


x = torch.randn(*shape, device=DEV)
full = build()
halves = [build() for _ in range(parts)]
xs = list(x.chunk(parts, 0))
streams = [torch.cuda.Stream() for _ in range(parts)]
want = full(x)
for m, xi in zip(halves, xs):
    m(xi)
torch.cuda.synchronize()


def run_full():
    return full(x)


def run_split():
    cur = torch.cuda.current_stream()
    outs = []
    for m, xi, st in zip(halves, xs, streams):
        st.wait_stream(cur)
        with torch.cuda.stream(st):
            outs.append(m(xi))
    for st in streams:
        cur.wait_stream(st)
    return torch.cat(outs, 0)


got = run_split()
torch.cuda.synchronize()
print("max |split - full| =", float((got - want).abs().max()))
for name, fn in (("full batch, one stream", run_full), ("%d part(s), %d streams" % (parts, parts), run_split), ("full batch, one stream", run_full)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print("%-28s %.3f ms  %.1f clips/s" % (name, ms, shape[0] * 1e3 / ms))
