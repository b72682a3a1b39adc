"""Conv programs (ptx_conv_program_*, csrc/conv_program.hip): several convolutions as ONE persistent launch.

No-GPU half: the planner is host code -- stage dependencies, queue sizes and refusals are checked with fake addresses.
GPU half: a program must be BIT-IDENTICAL to the same convs launched one by one on the same tile / split (same tile body,
same k-order, split-K partials summed in split order), and within fp32 tolerance of the ATen ops the reference calls
(resnet3D.py:125-143 bottleneck with shortcut B :175-185; r2plus1d.py:68-88 factored convs)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

DEV = "cuda:0"


def _r4(v):
    return (v + 3) // 4 * 4


def _desc(L, N, T, H, W, Ci, Co, k, s, p, relu=True, res=False, ldx=None, ldy=None):
    d = L.ConvDesc()
    To, Ho, Wo = ((T + 2 * p[0] - k[0]) // s[0] + 1, (H + 2 * p[1] - k[1]) // s[1] + 1, (W + 2 * p[2] - k[2]) // s[2] + 1)
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, ldx or _r4(Ci)
    d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, Co, ldy or _r4(Co)
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = (*k, *s, *p)
    d.Kc, d.Co_pad = _r4(Ci), (Co + 127) // 128 * 128
    d.flags = (L.PTX_EPI_RELU if relu else 0) | (L.PTX_EPI_RES_ADD if res else 0)
    if res:
        d.ldr = d.ldy
    return d


def _stage(L, d, x, w, b, y, res=0, x2=0, tile=-1, split=0):
    st = L.ConvStage()
    C.memmove(C.byref(st.desc), C.byref(d), C.sizeof(L.ConvDesc))
    st.x, st.x2, st.w_packed, st.bias, st.res, st.y = x, x2 or None, w, b, res or None, y
    st.tile, st.split_k = tile, split
    return st


def _arr(L, stages):
    a = (L.ConvStage * len(stages))()
    for i, s in enumerate(stages):
        C.memmove(C.byref(a[i]), C.byref(s), C.sizeof(L.ConvStage))
    return a


def _describe(L, lib, arr):
    """Plan as text: line 0 the totals, one "stage ..." line per stage, a last "queue <stage>.<group> ..." line."""
    buf = C.create_string_buffer(1 << 16)
    L.check(lib.ptx_conv_program_describe(arr, len(arr), buf, len(buf)), "describe")
    return buf.value.decode().splitlines()


def conv_config_for(lib, tile_name):
    """The ptx_conv3d_fwd tile configuration that computes what a program tile computes: same BM x BN x BK / waves / MFMA
    shape, hence the same k-order and the same split-K boundaries -- the LDS ring depth (dma / dma3 / dma4) does not change
    a single rounding (tests/test_gpu_kernels.py: test_conv_dma_bit_exact_vs_register_staged)."""
    names = {lib.ptx_conv3d_config_name(i).decode(): i for i in range(lib.ptx_conv3d_num_configs())}
    shape = "/".join(tile_name.split("/")[:3])
    for cand in (tile_name, shape + "/dma/re", shape + "/dma"):
        if cand in names:
            return names[cand]
    raise KeyError(tile_name)


# ------------------------------------------------------------------------------------------ host-only planner tests
def _fake_bottleneck(L, base=0x10000000):
    """conv1 -> conv2 (3x3x3) -> conv3 (+ residual = program input) with fake, disjoint, aligned addresses."""
    N, T, H, W = 2, 2, 7, 7
    A = [base + i * 0x1000000 for i in range(12)]
    d1 = _desc(L, N, T, H, W, 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    d2 = _desc(L, N, T, H, W, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    d3 = _desc(L, N, T, H, W, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), res=True)
    return [_stage(L, d1, A[0], A[4], A[5], A[1]), _stage(L, d2, A[1], A[6], A[7], A[2]),
            _stage(L, d3, A[2], A[8], A[9], A[3], res=A[0])], A


def test_planner_dependencies_and_sizes(ptx):
    L, lib = ptx._lib, ptx._lib.lib()
    stages, A = _fake_bottleneck(L)
    arr = _arr(L, stages)
    info = L.ConvProgramInfo()
    L.check(lib.ptx_conv_program_plan(arr, 3, C.byref(info)), "plan")
    lines = _describe(L, lib, arr)
    assert info.n_stages == 3 and info.total_items > 0 and info.ctrl_words % 64 == 0 and info.image_bytes % 16 == 0
    assert lines[1].split("deps")[1].strip() == ""             # conv1 reads the program's input only
    assert lines[2].split("deps")[1].split() == ["0:x"]        # conv2 <- conv1
    assert lines[3].split("deps")[1].split() == ["1:x"]        # conv3 <- conv2; its residual is external
    # halo of the 3x3x3 stage: (1*7 + 1)*7 + 1 rows either side
    assert " halo 57 57 " in lines[2]
    # queue length = sum over stages of tiles x splits, as described per stage
    assert info.total_items == sum(int(l.split(" items ")[1].split()[0]) for l in lines[1:] if l.startswith("stage"))
    # a second block whose residual IS stage 2's output: a "res" dependency on top of the input one
    d4 = _desc(L, 2, 2, 7, 7, 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    d5 = _desc(L, 2, 2, 7, 7, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), res=True)
    more = stages + [_stage(L, d4, A[3], A[4], A[5], A[10]), _stage(L, d5, A[10], A[8], A[9], A[11], res=A[3])]
    lines = _describe(L, lib, _arr(L, more))
    assert lines[4].split("deps")[1].split() == ["2:x"]
    assert sorted(lines[5].split("deps")[1].split()) == ["2:res", "3:x"]


def test_planner_wavefront_queue(ptx, monkeypatch):
    """The queue is a topological order of (stage, clip group) chunks: diagonal stage + group ascending, higher group first
    within a diagonal; PTX_PROG_GROUPS=1 gives plain stage order; every stage's tiles are queued exactly once."""
    L, lib = ptx._lib, ptx._lib.lib()
    N, T, H, W = 8, 2, 7, 7                      # 98 rows per clip: 32-row tiles straddle clips
    A = [0x10000000 + i * 0x1000000 for i in range(12)]
    d1 = _desc(L, N, T, H, W, 256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    d2 = _desc(L, N, T, H, W, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    d3 = _desc(L, N, T, H, W, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), res=True)
    arr = _arr(L, [_stage(L, d1, A[0], A[4], A[5], A[1]), _stage(L, d2, A[1], A[6], A[7], A[2]),
                   _stage(L, d3, A[2], A[8], A[9], A[3], res=A[0])])
    lines = _describe(L, lib, arr)
    head = dict(zip(lines[0].split()[0::2], lines[0].split()[1::2]))
    groups = int(head["groups"])
    assert groups == 4 and int(head["clips_per_group"]) == 2            # 98-row clips: two per group hold a 112-row tile
    q = [tuple(int(v) for v in c.split(".")) for c in lines[-1].split()[1:]]
    assert len(q) == int(head["chunks"]) == 3 * groups and len(set(q)) == len(q)
    diag = [s + g for s, g in q]
    assert diag == sorted(diag)
    for a, b in zip(q, q[1:]):
        if a[0] + a[1] == b[0] + b[1]:
            assert a[1] > b[1]
    info = L.ConvProgramInfo()
    L.check(lib.ptx_conv_program_plan(arr, 3, C.byref(info)), "plan")
    assert info.n_chunks == len(q)
    monkeypatch.setenv("PTX_PROG_GROUPS", "1")
    lines = _describe(L, lib, arr)
    assert lines[-1].split()[1:] == ["0.0", "1.0", "2.0"]


def test_planner_refusals(ptx):
    L, lib = ptx._lib, ptx._lib.lib()
    stages, A = _fake_bottleneck(L)
    info = L.ConvProgramInfo()
    # buffer reuse: stage 2 writes what stage 0 read
    bad = list(stages)
    bad[2] = _stage(L, bad[2].desc, A[2], A[8], A[9], A[0], res=A[3])
    assert lib.ptx_conv_program_plan(_arr(L, bad), 3, C.byref(info)) == 2
    assert b"own output" in lib.ptx_last_error()
    # shortcut-A residual / fp16 operands / grouped convs keep their own launches
    for flag in (L.PTX_EPI_RES_PADA, L.PTX_F16_OPERANDS, L.PTX_F16X3_OPERANDS):
        bad = list(stages)
        d = L.ConvDesc()
        C.memmove(C.byref(d), C.byref(bad[1].desc), C.sizeof(L.ConvDesc))
        d.flags |= flag
        bad[1] = _stage(L, d, A[1], A[6], A[7], A[2])
        assert lib.ptx_conv_program_plan(_arr(L, bad), 3, C.byref(info)) == 2
    # a consumer that reads the producer's rows through another row stride
    bad = list(stages)
    d = L.ConvDesc()
    C.memmove(C.byref(d), C.byref(bad[1].desc), C.sizeof(L.ConvDesc))
    d.ldx = 128
    bad[1] = _stage(L, d, A[1], A[6], A[7], A[2])
    assert lib.ptx_conv_program_plan(_arr(L, bad), 3, C.byref(info)) == 2
    assert lib.ptx_conv_program_plan(_arr(L, stages), 0, C.byref(info)) == 1


def test_planner_tile_names_are_conv_tiles(ptx):
    """Every tile shape of the program kernel is a tile configuration of ptx_conv3d_fwd: the bit-exactness contract names it."""
    lib = ptx._lib.lib()
    names = [lib.ptx_conv_program_tile_name(i).decode() for i in range(lib.ptx_conv_program_num_tiles())]
    assert names and len(set(names)) == len(names)
    for n in names:
        assert lib.ptx_conv3d_config_name(conv_config_for(lib, n)).decode().split("/")[:3] == n.split("/")[:3]


# ------------------------------------------------------------------------------------------ GPU parity
gpu = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


class Net:
    """A list of convs on device tensors: packs filters (BN folded), allocates every activation, and runs the list stage by
    stage (ptx_conv3d_fwd / _dual_fwd), as a program, or on the CPU with ATen."""

    def __init__(self, ptx, x):
        self.ptx, self.L, self.lib = ptx, ptx._lib, ptx._lib.lib()
        self.x_cpu = x
        N, Ci = x.shape[:2]
        xd = torch.zeros(N, *x.shape[2:], _r4(Ci))
        xd[..., :Ci] = x.permute(0, 2, 3, 4, 1)
        self.acts = [xd.to(DEV)]                  # channels-last device tensors; index 0 = the input
        self.cpu = [x]
        self.geo = [(N,) + tuple(x.shape[2:]) + (Ci,)]
        self.stages, self.keep, self.specs = [], [], []

    def conv(self, src, Co, k, s, p, seed, relu=True, res=None, x2=None, x2_stride=1):
        """Append conv(acts[src]) [+ conv1x1(acts[x2], stride) as a K-concatenated second source] [+ acts[res]]; returns its index."""
        L, lib = self.L, self.lib
        N, T, H, W, Ci = self.geo[src]
        w = _rnd(Co, Ci, *k, seed=seed, scale=(Ci * k[0] * k[1] * k[2]) ** -0.5)
        g = torch.Generator().manual_seed(seed + 1000)
        bn = (torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1, torch.randn(Co, generator=g) * 0.1,
              torch.rand(Co, generator=g) + 0.5, 1e-5)
        d = _desc(L, N, T, H, W, Ci, Co, k, s, p, relu=relu, res=res is not None)
        null = C.c_void_p(0)
        w2 = None
        if x2 is None:
            pd = L.PackDesc(Co, Ci, *k, _r4(Ci), d.Co_pad, 0)
            wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
            bp = torch.empty(pd.Co_pad, device=DEV)
            ts = [t.contiguous().to(DEV) for t in bn[:4]]
            wd = w.contiguous().to(DEV)
            L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), null, *[_p(t) for t in ts], C.c_float(bn[4]), _p(wp), _p(bp), _st()), "pack")
        else:
            # second source: a strided 1x1x1 conv of acts[x2] with its own BN, K-concatenated (ptx_pack_desc.ld_k / k_off)
            C2 = self.geo[x2][4]
            w2 = _rnd(Co, C2, 1, 1, 1, seed=seed + 1, scale=C2 ** -0.5)
            g2 = torch.Generator().manual_seed(seed + 2000)
            bn2 = (torch.rand(Co, generator=g2) + 0.5, torch.randn(Co, generator=g2) * 0.1, torch.randn(Co, generator=g2) * 0.1,
                   torch.rand(Co, generator=g2) + 0.5, 1e-5)
            ldk = _r4(Ci) + _r4(C2)
            wp = torch.zeros(d.Co_pad * ldk, device=DEV)
            bp = torch.zeros(d.Co_pad, device=DEV)
            for (ww, bb, cin, koff, acc) in ((w, bn, Ci, 0, 0), (w2, bn2, C2, _r4(Ci), 1)):
                pd = L.PackDesc(Co, cin, 1, 1, 1, _r4(cin), d.Co_pad, 0)
                pd.ld_k, pd.k_off, pd.bias_accumulate = ldk, koff, acc
                ts = [t.contiguous().to(DEV) for t in bb[:4]]
                wd = ww.contiguous().to(DEV)
                L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), null, *[_p(t) for t in ts], C.c_float(bb[4]), _p(wp), _p(bp), _st()), "pack dual")
                self.keep += ts + [wd]
            _, T2, H2, W2, _ = self.geo[x2]
            d.x2_C, d.x2_ld, d.x2_T, d.x2_H, d.x2_W = C2, _r4(C2), T2, H2, W2
            d.x2_sT = d.x2_sH = d.x2_sW = x2_stride
            bn = (bn, bn2)
        torch.cuda.synchronize()
        y = torch.full((N, d.To, d.Ho, d.Wo, d.ldy), float("nan"), device=DEV)
        self.acts.append(y)
        self.geo.append((N, d.To, d.Ho, d.Wo, Co))
        self.keep += [wp, bp]
        self.stages.append(dict(d=d, src=src, res=res, x2=x2, w=wp, b=bp, y=len(self.acts) - 1))
        self.specs.append(dict(src=src, w=w, w2=w2, bn=bn, s=s, p=p, relu=relu, res=res, x2=x2, x2_stride=x2_stride))
        return len(self.acts) - 1

    def reference(self):
        """ATen on the CPU, op by op as the reference modules issue them."""
        outs = [self.x_cpu]
        for sp in self.specs:
            bn = sp["bn"] if sp["x2"] is None else sp["bn"][0]
            y = F.batch_norm(F.conv3d(outs[sp["src"]], sp["w"], None, sp["s"], sp["p"]), bn[2], bn[3], bn[0], bn[1], False, 0.1, bn[4])
            if sp["x2"] is not None:
                b2 = sp["bn"][1]
                st = sp["x2_stride"]
                y = y + F.batch_norm(F.conv3d(outs[sp["x2"]], sp["w2"], None, (st, st, st)), b2[2], b2[3], b2[0], b2[1], False, 0.1, b2[4])
            if sp["res"] is not None:
                y = y + outs[sp["res"]]
            outs.append(F.relu(y) if sp["relu"] else y)
        return outs

    def clear(self):
        for a in self.acts[1:]:
            a.fill_(float("nan"))

    def stage_array(self, tiles=None, splits=None):
        L = self.L
        arr = (L.ConvStage * len(self.stages))()
        for i, s in enumerate(self.stages):
            e = arr[i]
            C.memmove(C.byref(e.desc), C.byref(s["d"]), C.sizeof(L.ConvDesc))
            e.x, e.w_packed, e.bias, e.y = _p(self.acts[s["src"]]), _p(s["w"]), _p(s["b"]), _p(self.acts[s["y"]])
            e.x2 = _p(self.acts[s["x2"]]) if s["x2"] is not None else None
            e.res = _p(self.acts[s["res"]]) if s["res"] is not None else None
            e.tile = -1 if tiles is None else tiles[i]
            e.split_k = 0 if splits is None else splits[i]
        return arr

    def run_program(self, wgs=2, tiles=None, splits=None, reps=1):
        L, lib = self.L, self.lib
        arr = self.stage_array(tiles, splits)
        info = L.ConvProgramInfo()
        L.check(lib.ptx_conv_program_plan(arr, len(arr), C.byref(info)), "plan")
        ws = torch.zeros(int(info.workspace_bytes) // 4 + 128, device=DEV)
        off = (-ws.data_ptr()) % 256 // 4
        ws = ws[off:]
        host = (C.c_char * int(info.image_bytes))()
        L.check(lib.ptx_conv_program_build(arr, len(arr), _p(ws), int(info.workspace_bytes), host, int(info.image_bytes), C.byref(info)), "build")
        image = torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(DEV)
        desc = _describe(L, lib, arr)
        outs = []
        for _ in range(reps):
            self.clear()
            L.check(lib.ptx_conv_program_fwd(C.byref(info), _p(image), _p(ws), wgs, _st()), "program")
            code = (C.c_int32 * 4)()
            L.check(lib.ptx_conv_program_error(_p(ws), code, _st()), "program error word")
            assert code[0] == 0, "program wait expired: %s\n%s" % (list(code), "\n".join(desc))
            outs.append([a.clone() for a in self.acts[1:]])
        return outs, desc, info

    def run_launches(self, desc_lines):
        """The same convs one launch each, on the tile / split the program's plan reports."""
        L, lib = self.L, self.lib
        self.clear()
        null = C.c_void_p(0)
        for s, line in zip(self.stages, desc_lines[1:]):
            f = line.split()
            cfg, split = conv_config_for(lib, f[3]), int(f[5])
            d = s["d"]
            nb = lib.ptx_conv3d_workspace_bytes(C.byref(d), max(split, 1))
            ws = torch.empty(max(nb // 4, 4), device=DEV)
            x, y = _p(self.acts[s["src"]]), _p(self.acts[s["y"]])
            if s["x2"] is not None:
                L.check(lib.ptx_conv3d_dual_fwd(C.byref(d), x, _p(self.acts[s["x2"]]), _p(s["w"]), _p(s["b"]), y, _p(ws), nb, cfg, split, _st()), "dual")
            else:
                r = _p(self.acts[s["res"]]) if s["res"] is not None else null
                L.check(lib.ptx_conv3d_fwd(C.byref(d), x, _p(s["w"]), _p(s["b"]), r, y, _p(ws), nb, cfg, split, _st()), "conv")
        torch.cuda.synchronize()
        return [a.clone() for a in self.acts[1:]]

    def to_ncdhw(self, i, t):
        Cc = self.geo[i][4]
        return t[..., :Cc].permute(0, 4, 1, 2, 3).contiguous().cpu()


def _bottlenecks(ptx, N=4, T=2, H=14, W=14, C0=256, planes=64, blocks=3, stride_first=True, seed=10, first_dual=True):
    """`blocks` ResNet3D bottlenecks: block 0 strided with shortcut B fused as a second K source (what the engine emits),
    the others with an identity residual.  resnet3D.py:125-143, :175-185."""
    net = Net(ptx, _rnd(N, C0, T, H, W, seed=seed))
    x = 0
    for b in range(blocks):
        s = (2, 2, 2) if (b == 0 and stride_first) else (1, 1, 1)
        o = net.conv(x, planes, (1, 1, 1), (1, 1, 1), (0, 0, 0), seed + 10 * b + 1)
        o = net.conv(o, planes, (3, 3, 3), s, (1, 1, 1), seed + 10 * b + 2)
        if b == 0 and first_dual:
            x = net.conv(o, planes * 4, (1, 1, 1), (1, 1, 1), (0, 0, 0), seed + 10 * b + 3, x2=x, x2_stride=s[0])
        else:
            x = net.conv(o, planes * 4, (1, 1, 1), (1, 1, 1), (0, 0, 0), seed + 10 * b + 3, res=x)
    return net


def _check(net, outs_prog, desc, tol=3e-4):
    want = net.reference()
    launches = net.run_launches(desc)
    for i, (a, b) in enumerate(zip(outs_prog, launches)):
        assert torch.equal(a, b), "stage %d: program differs from the launches on the same tile\n%s" % (i, "\n".join(desc))
    for i, a in enumerate(outs_prog):
        got, ref = net.to_ncdhw(i + 1, a), want[i + 1]
        err = (got - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), "stage %d: |d| %.3e vs ATen" % (i, err)
        pad = a[..., net.geo[i + 1][4]:]
        assert pad.numel() == 0 or bool((pad == 0).all())


@gpu
def test_program_resnet_bottlenecks_bit_exact(ptx):
    net = _bottlenecks(ptx)
    outs, desc, info = net.run_program()
    assert info.n_stages == 9 and any(" deps 5:res" in l or "5:res" in l.split("deps")[1] for l in desc[1:])
    _check(net, outs[0], desc)


@gpu
@pytest.mark.parametrize("wgs", [1, 2, 3])
def test_program_every_tile_split_and_grid(ptx, wgs):
    """Every tile shape x forced split-K, at 1-3 workgroups per CU; repeated launches must reproduce bit for bit (a missed
    dependency or a stale read shows up as a difference between repetitions or against the launches)."""
    lib = ptx._lib.lib()
    net = _bottlenecks(ptx, N=2, T=2, H=10, W=10, planes=64, blocks=2, seed=40)
    n = len(net.stages)
    for tile in range(lib.ptx_conv_program_num_tiles()):
        for split in (1, 3):
            outs, desc, _ = net.run_program(wgs=wgs, tiles=[tile] * n, splits=[split] * n, reps=3)
            for o in outs[1:]:
                assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
            _check(net, outs[0], desc)


@gpu
def test_program_2p1d_bottleneck_ragged_channels(ptx):
    """A (2+1)D bottleneck as the engine emits it: six GEMMs through mid widths that are no multiple of any tile
    (r2plus1d.py:68-69: 204 / 576 channels at p = 256), K tails and masked N tiles included."""
    net = Net(ptx, _rnd(3, 1024, 4, 7, 7, seed=70))
    x = 0
    for b in range(2):
        o = net.conv(x, 204, (1, 1, 1), (1, 1, 1), (0, 0, 0), 71 + 10 * b)
        o = net.conv(o, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), 72 + 10 * b)
        o = net.conv(o, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1), 73 + 10 * b)
        o = net.conv(o, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), 74 + 10 * b)
        o = net.conv(o, 204, (1, 1, 1), (1, 1, 1), (0, 0, 0), 75 + 10 * b)
        x = net.conv(o, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), 76 + 10 * b, res=x)
    outs, desc, info = net.run_program(reps=2)
    assert info.n_stages == 12
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    _check(net, outs[0], desc)


@gpu
def test_program_full_size_layer4_and_stress(ptx):
    """layer4 of resnet3d50 at BASELINE config 2 (M = 392, K up to 13824: deep split-K seams), 20 back-to-back launches."""
    net = _bottlenecks(ptx, N=8, T=2, H=14, W=14, C0=1024, planes=512, blocks=3, seed=90)
    outs, desc, _ = net.run_program(reps=20)
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(o, outs[0]))
    _check(net, outs[0], desc, tol=5e-4)


@gpu
def test_engine_runs_the_tail_as_programs(ptx, monkeypatch):
    """resnet3d50's layer3 / layer4 compile to conv programs; which of {program, its launches} runs is the tuner's call
    (PTX_PROGRAM=auto).  Forced either way the logits agree to fp32 summation-order noise (the program picks its own tiles /
    splits; same-tile bit-exactness is tested above), and the program's error word stays clear."""
    import pretorched_x_amd as P
    monkeypatch.setenv("PTX_PROGRAM", "force")       # (auto keeps a program only where the tuner measured it faster)
    torch.manual_seed(0)
    m = P.resnet3d50(num_classes=17, pretrained=None).eval().to(DEV)
    x = torch.randn(2, 3, 8, 112, 112, generator=torch.Generator().manual_seed(1)).to(DEV)
    with torch.no_grad():
        m(x)
        plan = list(m.engine()._plans.values())[-1]
        assert plan.program_steps
        n_prog = sum(len(p.convs) for p in plan.program_steps)
        assert n_prog >= 15, n_prog
        outs = []
        for flag in (True, False):
            for p in plan.program_steps:
                p.use_program = flag
            outs.append(m(x).clone())
            assert all(p.error() is None for p in plan.program_steps)
        launches = len(plan.all_convs())
        for p in plan.program_steps:
            p.use_program = True
        assert len(plan.all_convs()) == launches - n_prog + len(plan.program_steps)
    y1, y0 = outs
    assert (y0 - y1).abs().max().item() <= 1e-4 * max(1.0, y0.abs().max().item())


@gpu
def test_engine_programs_on_the_2p1d_nonlocal_composite_and_auto_mode(ptx, monkeypatch):
    """BASELINE config 3's network at a small input: forced programs ((2+1)D factored GEMMs through ragged widths, NL
    projections, attention launches between the programs) agree with the launch-per-conv plan; under PTX_PROGRAM=auto the
    tuner times both, records its choice ("prog:" keys of the tuned table) and dissolves the programs it rejects."""
    import pretorched_x_amd as P
    from pretorched_x_amd import engine as E
    x = torch.randn(2, 3, 16, 64, 64, generator=torch.Generator().manual_seed(5)).to(DEV)

    def run(mode):
        monkeypatch.setenv("PTX_PROGRAM", mode)
        torch.manual_seed(3)
        m = P.nonlocal_r2plus1d50(num_classes=21).eval().to(DEV)
        with torch.no_grad():
            y = m(x).clone()
        return y, list(m.engine()._plans.values())[-1]

    y0, plan0 = run("0")
    assert not plan0.program_steps
    y1, plan1 = run("force")
    assert plan1.program_steps and all(p.use_program and p.error() is None for p in plan1.program_steps)
    assert len(plan1.all_convs()) < len(plan0.all_convs())
    assert (y0 - y1).abs().max().item() <= 1e-4 * max(1.0, y0.abs().max().item())
    before = {k for k in E.tuned_snapshot() if k.startswith("prog:")}
    y2, plan2 = run("auto")
    after = {k: v for k, v in E.tuned_snapshot().items() if k.startswith("prog:")}
    assert len(after) > len(before)                                  # every run of convs got a measured verdict
    assert all(p.use_program for p in plan2.program_steps)           # what is still a program won its A/B
    assert (y0 - y2).abs().max().item() <= 1e-4 * max(1.0, y0.abs().max().item())
