"""Visibility probe for conv programs (csrc/conv_program.hip): runs progressively larger programs and WATCHES the
program's control words (queue head, error word, completion counters) from a side stream while the launch is in flight, so
a launch that does not finish reports where it stopped instead of timing out blind.  Exits (os._exit) after `limit` seconds.

    python scripts/gpu_prog_probe.py [case ...]      cases: plain one two split block full layer4
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
torch.set_num_threads(16)

import pretorched_x_amd as ptx  # noqa: E402
import test_conv_program as T  # noqa: E402

DEV = "cuda:0"
L, lib = ptx._lib, ptx._lib.lib()


def say(*a):
    print(*a, flush=True)


def watch_program(net, wgs=2, tiles=None, splits=None, limit=12.0):
    arr = net.stage_array(tiles, splits)
    info = L.ConvProgramInfo()
    L.check(lib.ptx_conv_program_plan(arr, len(arr), C.byref(info)), "plan")
    desc = T._describe(L, lib, arr)
    for line in desc:
        say("   ", line)
    ws = torch.zeros(int(info.workspace_bytes) // 4 + 128, device=DEV)
    ws = ws[(-ws.data_ptr()) % 256 // 4:]
    host = (C.c_char * int(info.image_bytes))()
    L.check(lib.ptx_conv_program_build(arr, len(arr), T._p(ws), int(info.workspace_bytes), host, int(info.image_bytes), C.byref(info)), "build")
    image = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(DEV)
    net.clear()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    done = torch.cuda.Event()
    t0 = time.time()
    L.check(lib.ptx_conv_program_fwd(C.byref(info), T._p(image), T._p(ws), wgs, T._st()), "program")
    done.record()
    nctrl = int(info.ctrl_words)
    last = None
    while not done.query():
        with torch.cuda.stream(side):
            snap = ws[:nctrl].view(torch.int32).to("cpu", non_blocking=False)
        cur = (int(snap[0]), int(snap[1]), int(snap[16:].sum()))
        if cur != last:
            say("    t=%.2fs head=%d/%d err=%s done_sum=%d" % (time.time() - t0, cur[0], info.total_items, list(map(int, snap[1:5])), cur[2]))
            last = cur
        if time.time() - t0 > limit:
            say("    NOT FINISHED after %.1fs: ctrl[0:8]=%s" % (limit, list(map(int, snap[:8]))))
            say("    counters:", list(map(int, snap[16:16 + 96])))
            sys.stdout.flush()
            os._exit(3)
        time.sleep(0.05)
    torch.cuda.synchronize()
    say("    finished in %.3fs (host wall, first launch)" % (time.time() - t0))
    code = (C.c_int32 * 4)()
    L.check(lib.ptx_conv_program_error(T._p(ws), code, T._st()), "err")
    say("    error word:", list(code))
    outs = [a.clone() for a in net.acts[1:]]
    want = net.reference()
    launches = net.run_launches(desc)
    for i, (a, b) in enumerate(zip(outs, launches)):
        eq = bool(torch.equal(a, b))
        got, ref = net.to_ncdhw(i + 1, a), want[i + 1]
        err = (got - ref).abs().max().item()
        lerr = (net.to_ncdhw(i + 1, b) - ref).abs().max().item()
        say("    stage %d: program==launches %s   |program-aten| %.3e   |launches-aten| %.3e   (max |ref| %.2f)" % (i, eq, err, lerr, ref.abs().max().item()))
    # timing: 20 back-to-back launches vs the launches
    for name, fn in (("program", lambda: L.check(lib.ptx_conv_program_fwd(C.byref(info), T._p(image), T._p(ws), wgs, T._st()), "p")),):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        e1.synchronize()
        say("    %s: %.1f us per launch (wgs %d)" % (name, 1e3 * e0.elapsed_time(e1) / 20, wgs))


def build_program(net, tiles=None, splits=None):
    arr = net.stage_array(tiles, splits)
    info = L.ConvProgramInfo()
    L.check(lib.ptx_conv_program_plan(arr, len(arr), C.byref(info)), "plan")
    ws = torch.zeros(int(info.workspace_bytes) // 4 + 128, device=DEV)
    ws = ws[(-ws.data_ptr()) % 256 // 4:]
    host = (C.c_char * int(info.image_bytes))()
    L.check(lib.ptx_conv_program_build(arr, len(arr), T._p(ws), int(info.workspace_bytes), host, int(info.image_bytes), C.byref(info)), "build")
    image = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(DEV)
    return arr, info, ws, image, T._describe(L, lib, arr)


def time_fn(fn, reps=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def launches_fn(net, desc):
    """closure running the stages one launch each on the tiles / splits the plan names"""
    null = C.c_void_p(0)
    calls = []
    keep = []
    for s, line in zip(net.stages, desc[1:]):
        f = line.split()
        cfg, split = T.conv_config_for(lib, f[3]), int(f[5])
        d = s["d"]
        nb = lib.ptx_conv3d_workspace_bytes(C.byref(d), max(split, 1))
        ws = torch.empty(max(nb // 4, 4), device=DEV)
        keep.append(ws)
        x, y = T._p(net.acts[s["src"]]), T._p(net.acts[s["y"]])
        if s["x2"] is not None:
            calls.append((lib.ptx_conv3d_dual_fwd, (C.byref(d), x, T._p(net.acts[s["x2"]]), T._p(s["w"]), T._p(s["b"]), y, T._p(ws), nb, cfg, split)))
        else:
            r = T._p(net.acts[s["res"]]) if s["res"] is not None else null
            calls.append((lib.ptx_conv3d_fwd, (C.byref(d), x, T._p(s["w"]), T._p(s["b"]), r, y, T._p(ws), nb, cfg, split)))

    def run():
        st = T._st()
        for f_, a in calls:
            L.check(f_(*a, st), "launch")
    run.keep = keep
    return run


def trace_program(net, wgs=2, tiles=None, splits=None, label=""):
    """One traced launch: where the workgroups' time goes, per stage."""
    arr, info, ws, image, desc = build_program(net, tiles, splits)
    n = int(info.total_items)
    fwd = lambda: L.check(lib.ptx_conv_program_fwd(C.byref(info), T._p(image), T._p(ws), wgs, T._st()), "p")   # noqa: E731
    us = time_fn(fwd)
    lus = time_fn(launches_fn(net, desc))
    trace = torch.zeros(n * 8, dtype=torch.int64, device=DEV)
    L.check(lib.ptx_conv_program_trace_fwd(C.byref(info), T._p(image), T._p(ws), wgs, C.c_void_p(trace.data_ptr()), n * 64, T._st()), "trace")
    torch.cuda.synchronize()
    t = trace.cpu().view(n, 8)
    t0, t1, t2, t3 = [t[:, i].double() / 100.0 for i in range(4)]          # us
    stage = (t[:, 5] & 0xffffffff).long()
    base = t0.min()
    span = (t3.max() - base).item()
    grid = min(wgs * 256, n)
    say("  %s program %.1f us | same tiles as launches %.1f us | traced span %.1f us, %d items on %d workgroups (%d per CU)" % (
        label, us, lus, span, n, grid, wgs))
    say("   stage tile                      split items   start    end   | per item: take->go   tile  publish (us) | sum tile us")
    for sidx in range(int(info.n_stages)):
        m = stage == sidx
        f = desc[1 + sidx].split()
        say("   %5d %-26s %5s %5d %7.1f %7.1f |           %7.2f %7.2f %7.2f        | %9.1f" % (
            sidx, f[3], f[5], int(m.sum()), (t0[m].min() - base).item(), (t3[m].max() - base).item(),
            (t1[m] - t0[m]).mean().item(), (t2[m] - t1[m]).mean().item(), (t3[m] - t2[m]).mean().item(), (t2[m] - t1[m]).sum().item()))
    busy = (t2 - t1).sum().item()
    wait = (t1 - t0).sum().item()
    pub = (t3 - t2).sum().item()
    say("   workgroup-time: tile %.0f us (%.0f%% of %d x span), waiting %.0f us (%.0f%%), publish/reduce %.0f us (%.0f%%)" % (
        busy, 100 * busy / (grid * span), grid, wait, 100 * wait / (grid * span), pub, 100 * pub / (grid * span)))
    return us, lus


def sweep(net, label):
    say("== sweep %s" % label)
    for target in (256, 512, 768):
        for min_steps in (8, 16, 32):
            os.environ["PTX_PROG_TARGET_ITEMS"], os.environ["PTX_PROG_MIN_STEPS"] = str(target), str(min_steps)
            arr, info, ws, image, desc = build_program(net)
            row = []
            for wgs in (1, 2, 3):
                fwd = lambda: L.check(lib.ptx_conv_program_fwd(C.byref(info), T._p(image), T._p(ws), wgs, T._st()), "p")   # noqa: E731
                row.append(time_fn(fwd, 20))
            lus = time_fn(launches_fn(net, desc), 20)
            say("   target %4d min_steps %2d: items %5d  program wgs1/2/3 %7.1f %7.1f %7.1f us | launches (same tiles) %7.1f us   splits %s" % (
                target, min_steps, info.total_items, row[0], row[1], row[2], lus, [int(l.split()[5]) for l in desc[1:]]))
    os.environ.pop("PTX_PROG_TARGET_ITEMS", None)
    os.environ.pop("PTX_PROG_MIN_STEPS", None)


def tune(net, kinds, label, rounds=2):
    """Coordinate descent over (tile, split) per stage KIND (stages of the same kind share the choice), program time as the
    objective; then the groups / workgroups-per-CU grid on the winner.  Prints every improvement."""
    say("== tune %s" % label)
    ntiles = lib.ptx_conv_program_num_tiles()
    names = [lib.ptx_conv_program_tile_name(i).decode() for i in range(ntiles)]
    nk = max(kinds) + 1
    choice = [(-1, 0)] * nk

    def measure(ch, wgs_list=(2, 3), reps=10):
        tiles = [ch[k][0] for k in kinds]
        splits = [ch[k][1] for k in kinds]
        try:
            arr, info, ws, image, desc = build_program(net, tiles, splits)
        except Exception as e:      # noqa: BLE001
            return None, None, str(e)
        best = None
        for wgs in wgs_list:
            fwd = lambda: L.check(lib.ptx_conv_program_fwd(C.byref(info), T._p(image), T._p(ws), wgs, T._st()), "p")   # noqa: E731
            t = time_fn(fwd, reps)
            if best is None or t < best[0]:
                best = (t, wgs)
        return best[0], best[1], desc

    cur, wg, desc = measure(choice)
    say("   start (library defaults): %.1f us (wgs %d)" % (cur, wg))
    for r in range(rounds):
        for k in range(nk):
            for tile in range(ntiles):
                for split in (1, 2, 3, 4, 6, 8):
                    trial = list(choice)
                    trial[k] = (tile, split)
                    t, w, _ = measure(trial)
                    if t is not None and t < cur * 0.995:
                        cur, wg, choice = t, w, trial
                        say("   round %d kind %d -> %-28s split %d : %.1f us (wgs %d)" % (r, k, names[tile], split, t, w))
    say("   best per kind: %s" % [(names[t] if t >= 0 else "auto", sp) for t, sp in choice])
    for groups in ("1", "2", "4", "8"):
        os.environ["PTX_PROG_GROUPS"] = groups
        row = []
        for wgs in (1, 2, 3):
            t, _, d = measure(choice, (wgs,), 20)
            row.append(t)
        say("   groups %s: wgs1/2/3 %7.1f %7.1f %7.1f us   %s" % (groups, row[0], row[1], row[2], d[0]))
    os.environ.pop("PTX_PROG_GROUPS", None)
    tiles = [choice[k][0] for k in kinds]
    splits = [choice[k][1] for k in kinds]
    arr, info, ws, image, desc = build_program(net, tiles, splits)
    say("   launches on the same tiles / splits: %.1f us" % time_fn(launches_fn(net, desc), 20))
    for wgs in (2, 3):
        trace_program(net, wgs, tiles, splits, label=label + " tuned")
    return choice


def auto_launches(net, stream_ptr_fn):
    """the stages one launch each on the library's own tile pick (what a plan without a tuned-table hit runs)"""
    null = C.c_void_p(0)
    calls, keep = [], []
    for s in net.stages:
        d = s["d"]
        sk = C.c_int(1)
        cfg = lib.ptx_conv3d_pick_config(C.byref(d), C.byref(sk))
        nb = lib.ptx_conv3d_workspace_bytes(C.byref(d), max(sk.value, 1))
        ws = torch.empty(max(nb // 4, 4), device=DEV)
        keep.append(ws)
        x, y = T._p(net.acts[s["src"]]), T._p(net.acts[s["y"]])
        if s["x2"] is not None:
            calls.append((lib.ptx_conv3d_dual_fwd, (C.byref(d), x, T._p(net.acts[s["x2"]]), T._p(s["w"]), T._p(s["b"]), y, T._p(ws), nb, cfg, sk.value)))
        else:
            r = T._p(net.acts[s["res"]]) if s["res"] is not None else null
            calls.append((lib.ptx_conv3d_fwd, (C.byref(d), x, T._p(s["w"]), T._p(s["b"]), r, y, T._p(ws), nb, cfg, sk.value)))

    def run():
        st = stream_ptr_fn()
        for f_, a in calls:
            L.check(f_(*a, st), "launch")
    run.keep = keep
    return run


def streams_case(make_net, label):
    """One full-batch chain on one stream vs the same work as two half-batch chains on two streams (clips are independent):
    does the hardware's own scheduler overlap the small launches of different stages?"""
    say("== streams %s" % label)
    full = make_net(8)
    halves = [make_net(4), make_net(4)]
    main = torch.cuda.current_stream()
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    f_full = auto_launches(full, lambda: C.c_void_p(main.cuda_stream))
    f_half = [auto_launches(h, (lambda s_: (lambda: C.c_void_p(s_.cuda_stream)))(s_)) for h, s_ in zip(halves, side)]
    ev0, ev = torch.cuda.Event(), [torch.cuda.Event(), torch.cuda.Event()]

    def two_streams():
        ev0.record(main)
        for i in range(2):
            side[i].wait_event(ev0)
            f_half[i]()
            ev[i].record(side[i])
        for i in range(2):
            main.wait_event(ev[i])

    def one_stream_halves():
        for i in range(2):
            auto_main[i]()
    auto_main = [auto_launches(h, lambda: C.c_void_p(main.cuda_stream)) for h in halves]
    t_full = time_fn(f_full, 30)
    t_two = time_fn(two_streams, 30)
    t_seq = time_fn(one_stream_halves, 30)
    say("   full batch, one stream: %.1f us | two half-batch chains on two streams: %.1f us (%.2fx) | the halves back to back on one stream: %.1f us" % (
        t_full, t_two, t_full / t_two, t_seq))


def nets(name):
    if name == "layer3":       # resnet3d50 layer3.1-3.3 at config 2: M = 3136, identity blocks only
        return T._bottlenecks(ptx, N=8, T=2, H=14, W=14, C0=1024, planes=256, blocks=3, stride_first=False, seed=90, first_dual=False)
    if name == "layer4":
        return T._bottlenecks(ptx, N=8, T=1, H=7, W=7, C0=2048, planes=512, blocks=3, stride_first=False, seed=91, first_dual=False)
    if name == "2p1d":         # config 3 layer3.1-3.2: (2+1)D bottlenecks at M = 1568
        net = T.Net(ptx, T._rnd(8, 1024, 4, 7, 7, seed=70))
        x = 0
        for b in range(2):
            o = net.conv(x, 204, (1, 1, 1), (1, 1, 1), (0, 0, 0), 71 + 10 * b)
            o = net.conv(o, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), 72 + 10 * b)
            o = net.conv(o, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1), 73 + 10 * b)
            o = net.conv(o, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 74 + 10 * b)
            o = net.conv(o, 204, (1, 1, 1), (1, 1, 1), (0, 0, 0), 75 + 10 * b)
            x = net.conv(o, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), 76 + 10 * b, res=x)
        return net
    raise SystemExit(name)


def plain_sanity():
    """The refactored tile body as a plain launch (library default tile) against ATen."""
    net = T._bottlenecks(ptx, N=2, T=2, H=10, W=10, planes=64, blocks=1, seed=5)
    net.clear()
    null = C.c_void_p(0)
    for s in net.stages:
        d = s["d"]
        nb = lib.ptx_conv3d_workspace_bytes(C.byref(d), 8)
        ws = torch.empty(max(nb // 4, 4), device=DEV)
        x, y = T._p(net.acts[s["src"]]), T._p(net.acts[s["y"]])
        if s["x2"] is not None:
            L.check(lib.ptx_conv3d_dual_fwd(C.byref(d), x, T._p(net.acts[s["x2"]]), T._p(s["w"]), T._p(s["b"]), y, T._p(ws), nb, -1, 0, T._st()), "dual")
        else:
            L.check(lib.ptx_conv3d_fwd(C.byref(d), x, T._p(s["w"]), T._p(s["b"]), null, y, T._p(ws), nb, -1, 0, T._st()), "conv")
    torch.cuda.synchronize()
    want = net.reference()
    for i, a in enumerate(net.acts[1:]):
        say("    plain stage %d: |d| %.3e" % (i, (net.to_ncdhw(i + 1, a) - want[i + 1]).abs().max().item()))


def case(name):
    say("== %s" % name)
    if name == "plain":
        return plain_sanity()
    if name == "one":          # one pointwise stage, no dependencies
        net = T.Net(ptx, T._rnd(2, 64, 2, 8, 8, seed=1))
        net.conv(0, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), 2)
        return watch_program(net, splits=[1])
    if name == "two":          # two stages, one dependency
        net = T.Net(ptx, T._rnd(2, 64, 2, 8, 8, seed=1))
        o = net.conv(0, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), 2)
        net.conv(o, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 3)
        return watch_program(net, splits=[1, 1])
    if name == "split":        # the same with split-K seams
        net = T.Net(ptx, T._rnd(2, 64, 2, 8, 8, seed=1))
        o = net.conv(0, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), 2)
        net.conv(o, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1), 3)
        return watch_program(net, splits=[2, 3])
    if name == "block":
        return watch_program(T._bottlenecks(ptx, N=2, T=2, H=10, W=10, planes=64, blocks=2, seed=40))
    if name == "full":
        return watch_program(T._bottlenecks(ptx))
    if name == "layer3":       # resnet3d50 layer3 at config 2: M = 3136
        return watch_program(T._bottlenecks(ptx, N=8, T=4, H=28, W=28, C0=512, planes=256, blocks=3, seed=90), limit=20)
    if name == "layer4":
        return watch_program(T._bottlenecks(ptx, N=8, T=2, H=14, W=14, C0=1024, planes=512, blocks=3, seed=90), limit=20)
    if name.startswith("trace:"):
        net = nets(name[6:])
        for wgs in (1, 2, 3):
            trace_program(net, wgs, label=name[6:])
        return
    if name.startswith("sweep:"):
        return sweep(nets(name[6:]), name[6:])
    if name.startswith("streams:"):
        kind = name[8:]
        if kind == "layer3":
            mk = lambda n: T._bottlenecks(ptx, N=n, T=2, H=14, W=14, C0=1024, planes=256, blocks=3, stride_first=False, seed=90, first_dual=False)   # noqa: E731
        elif kind == "layer4":
            mk = lambda n: T._bottlenecks(ptx, N=n, T=1, H=7, W=7, C0=2048, planes=512, blocks=3, stride_first=False, seed=91, first_dual=False)   # noqa: E731
        else:
            def mk(n):
                net = T.Net(ptx, T._rnd(n, 1024, 4, 7, 7, seed=70))
                x = 0
                for b in range(2):
                    o = net.conv(x, 204, (1, 1, 1), (1, 1, 1), (0, 0, 0), 71 + 10 * b)
                    o = net.conv(o, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), 72 + 10 * b)
                    o = net.conv(o, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1), 73 + 10 * b)
                    o = net.conv(o, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 74 + 10 * b)
                    o = net.conv(o, 204, (1, 1, 1), (1, 1, 1), (0, 0, 0), 75 + 10 * b)
                    x = net.conv(o, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), 76 + 10 * b, res=x)
                return net
        return streams_case(mk, kind)
    if name.startswith("tune:"):
        net = nets(name[5:])
        per_block = 6 if name[5:] == "2p1d" else 3
        return tune(net, [i % per_block for i in range(len(net.stages))], name[5:])
    raise SystemExit("unknown case " + name)


if __name__ == "__main__":
    say(torch.cuda.get_device_name(0), lib.ptx_version().decode())
    for c in (sys.argv[1:] or ["plain", "one", "two", "split", "block", "full"]):
        case(c)
    say("probe done")
