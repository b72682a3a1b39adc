"""GPU parity tests, kernel level: every C-ABI entry point against the ATen CPU op the reference
calls at the cited site, on seeded inputs, through ctypes (no torch GPU math in the path under
test -- torch only moves bytes).  Tolerances: fp32 with a different summation order."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _lib(ptx):
    return ptx._lib.lib()


def _p(t, off=0):
    return C.c_void_p(t.data_ptr() + 4 * off)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _r4(v):
    return (v + 3) // 4 * 4


def to_cl(x, ld=None):
    """NCDHW cpu -> channels-last [N,T,H,W,ld] on the GPU (zero pad channels)."""
    n, c = x.shape[:2]
    ld = _r4(c) if ld is None else ld
    y = torch.zeros(n, *x.shape[2:], ld)
    y[..., :c] = x.permute(0, 2, 3, 4, 1)
    return y.to(DEV)


def from_cl(y, c):
    return y[..., :c].permute(0, 4, 1, 2, 3).contiguous().cpu()


def hip_conv(ptx, x, w, stride, padding, bias=None, bn=None, relu=False, res=None, res_pad=None, res_stride=1,
             cfg=-1, split=0, pro_relu=False, x3=False, fused_split=None):
    """x NCDHW cpu, w [Co,Ci,kT,kH,kW] cpu.  Returns NCDHW cpu output of ptx_conv3d_fwd.
    x3: split operands (PTX_F16X3_OPERANDS) -- the filter packed as (hi8 | lo8) half blocks."""
    L, lib = ptx._lib, _lib(ptx)
    Co, Ci, kT, kH, kW = w.shape
    N, _, T, H, W = x.shape
    sT, sH, sW = stride
    pT, pH, pW = padding
    To, Ho, Wo = (T + 2 * pT - kT) // sT + 1, (H + 2 * pH - kH) // sH + 1, (W + 2 * pW - kW) // sW + 1
    pd = L.PackDesc(Co, Ci, kT, kH, kW, (Ci + 7) // 8 * 8 if x3 else _r4(Ci), (Co + 127) // 128 * 128, 0)
    pd.f16 = 2 if x3 else 0
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(pd.Co_pad, device=DEV)
    wd = w.contiguous().to(DEV)
    null = C.c_void_p(0)
    keep = [wd]
    bnargs, eps = [null] * 4, 0.0
    if bn is not None:
        ts = [t.contiguous().to(DEV) for t in bn[:4]]
        keep += ts
        bnargs, eps = [_p(t) for t in ts], bn[4]
    bd = bias.to(DEV) if bias is not None else None
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bd) if bd is not None else null, *bnargs,
                                     C.c_float(eps), _p(wp), _p(bp), _st()), "pack")
    xd = to_cl(x)
    ldy = _r4(Co)
    yd = torch.full((N, To, Ho, Wo, ldy), float("nan"), device=DEV)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, xd.shape[-1]
    d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, Co, ldy
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = kT, kH, kW, sT, sH, sW, pT, pH, pW
    d.Kc, d.Co_pad = pd.Kc, pd.Co_pad
    flags = (L.PTX_EPI_RELU if relu else 0) | (L.PTX_PRO_RELU if pro_relu else 0) | (L.PTX_F16X3_OPERANDS if x3 else 0)
    rd = None
    if res is not None:
        rd = to_cl(res)
        d.ldr = rd.shape[-1]
        flags |= L.PTX_EPI_RES_ADD
    if res_pad is not None:
        rd = to_cl(res_pad)
        d.ldr = rd.shape[-1]
        d.res_C, d.res_T, d.res_H, d.res_W = res_pad.shape[1], res_pad.shape[2], res_pad.shape[3], res_pad.shape[4]
        d.res_sT = d.res_sH = d.res_sW = res_stride
        flags |= L.PTX_EPI_RES_PADA
    if fused_split is not None:          # PTX_SPLITK_FUSED: `fused_split` is the caller's zeroed workspace (reused across calls)
        flags |= L.PTX_SPLITK_FUSED
    d.flags = flags
    ws_bytes = lib.ptx_conv3d_workspace_bytes(C.byref(d), 8)
    if fused_split is not None:
        assert fused_split.numel() * 4 >= ws_bytes
        ws = fused_split
    else:
        ws = torch.empty(max(ws_bytes // 4, 4), device=DEV)
    L.check(lib.ptx_conv3d_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), _p(rd) if rd is not None else null, _p(yd),
                               _p(ws), ws_bytes, cfg, split, _st()), "conv")
    torch.cuda.synchronize()
    pad = yd[..., Co:]
    assert pad.numel() == 0 or bool((pad == 0).all()), "pad channels must be written as zero"
    return from_cl(yd, Co)


def ref_conv(x, w, stride, padding, bias=None, bn=None, relu=False, res=None, res_pad=None, res_stride=1,
             pro_relu=False):
    if pro_relu:
        x = F.relu(x)
    y = F.conv3d(x, w, bias, stride, padding)
    if bn is not None:
        g, b, m, v, eps = bn
        y = F.batch_norm(y, m, v, g, b, False, 0.1, eps)
    if res is not None:
        y = y + res
    if res_pad is not None:
        r = F.avg_pool3d(res_pad, kernel_size=1, stride=res_stride)
        y = y + torch.cat([r, torch.zeros(r.size(0), y.size(1) - r.size(1), *r.shape[2:])], 1)
    return F.relu(y) if relu else y


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def make_bn(c, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1,
            torch.rand(c, generator=g) + 0.5, 1e-5)


def close(got, want, tol=2e-4):
    scale = max(1.0, want.abs().max().item())
    err = (got - want).abs().max().item()
    assert got.shape == want.shape
    assert err <= tol * scale, "max err %.3e (scale %.3e)" % (err, scale)


GEOMS = [
    # name, N,T,H,W, Ci,Co, k, s, p
    ("pw64_256", 2, 3, 10, 11, 64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("c3s1_64", 2, 3, 10, 11, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("c3s2_128", 1, 6, 14, 14, 128, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ("pw_s2", 2, 4, 14, 14, 256, 512, (1, 1, 1), (2, 2, 2), (0, 0, 0)),
    ("c3_T1", 3, 1, 7, 7, 128, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("c3_T2_s2", 2, 2, 14, 14, 64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    ("spatial_odd", 2, 4, 9, 9, 51, 85, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
    ("temporal_odd", 2, 5, 6, 6, 85, 34, (3, 1, 1), (2, 1, 1), (1, 0, 0)),
    ("temporal7", 1, 8, 6, 6, 110, 64, (7, 1, 1), (1, 1, 1), (3, 0, 0)),
    ("c5_generic", 1, 5, 9, 10, 12, 20, (3, 5, 2), (1, 2, 1), (1, 2, 1)),
]


@pytest.mark.parametrize("g", GEOMS, ids=[g[0] for g in GEOMS])
def test_conv_geometries_auto_config(ptx, g):
    name, N, T, H, W, Ci, Co, k, s, p = g
    x, w = rnd(N, Ci, T, H, W, seed=1), rnd(Co, Ci, *k, seed=2, scale=(Ci * k[0] * k[1] * k[2]) ** -0.5)
    bn = make_bn(Co, 3)
    close(hip_conv(ptx, x, w, s, p, bn=bn, relu=True), ref_conv(x, w, s, p, bn=bn, relu=True))


def fp32_configs(lib):
    """Tile configurations that take fp32 operands (the '/f16' and '/x3' ones have their own tests)."""
    return [i for i in range(lib.ptx_conv3d_num_configs())
            if not lib.ptx_conv3d_config_name(i).decode().endswith(("/f16", "/x3"))]


def x3_configs(lib):
    """Split-operand tiles that run any geometry (the kw-reuse ones need whole-row tiles: kwr_configs)."""
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    return [i for i, n in enumerate(names) if n.endswith("/x3") and "/kwr/" not in n]


def kwr_configs(lib, kind):
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    return [i for i, n in enumerate(names) if n.endswith("/kwr/" + kind)]


def test_conv_every_config_and_split(ptx):
    lib = _lib(ptx)
    N, T, H, W, Ci, Co = 2, 3, 9, 10, 64, 160
    x, w = rnd(N, Ci, T, H, W, seed=4), rnd(Co, Ci, 3, 3, 3, seed=5, scale=0.03)
    bn = make_bn(Co, 6)
    res = rnd(N, Co, T, H, W, seed=7)
    want = ref_conv(x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, res=res)
    errs = {}
    for cfg in fp32_configs(lib):
        for split in (1, 2, 5):
            got = hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, res=res, cfg=cfg, split=split)
            errs[(lib.ptx_conv3d_config_name(cfg).decode(), split)] = (got - want).abs().max().item()
    bad = {k: v for k, v in errs.items() if not v <= 2e-4 * max(1.0, want.abs().max().item())}
    assert not bad, bad


def test_conv_dma_bit_exact_vs_register_staged(ptx):
    """Race detector for the LDS-DMA pipeline: tiles with the same MFMA shape accumulate every output
    in the same k-order whatever the staging flavour or BK (16 | 32), so results must be bit-identical
    to the register-staged kernel -- over repeated launches on a problem big enough to fill the chip
    (any stale / early LDS read would show up as a difference)."""
    lib = _lib(ptx)
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    x, w = rnd(4, 64, 6, 40, 40, seed=50), rnd(128, 64, 3, 3, 3, seed=51, scale=0.03)
    bn = make_bn(128, 52)
    base = hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, cfg=names.index("64x64x32/2x2/m32"), split=1)
    for name in ("64x64x32/2x2/m32/dma", "64x64x16/2x2/m32/dma", "64x64x16/2x2/m32", "128x64x32/4x2/m32/dma",
                 "128x128x32/4x2/m32/dma", "64x128x16/2x2/m32/dma", "256x64x32/8x1/m32/dma",
                 "64x64x16/2x2/m32/dma3", "64x64x32/2x2/m32/dma3", "128x64x32/4x2/m32/dma3",
                 "128x128x32/4x2/m32/dma3", "64x128x16/2x2/m32/dma3", "128x64x16/2x2/m32/dma3",
                 "64x64x32/2x2/m32/dma4", "64x64x16/2x2/m32/dma4", "64x128x32/2x2/m32/dma4",
                 "64x64x64/2x2/m32/dma", "64x128x64/2x2/m32/dma", "128x64x64/4x2/m32/dma",
                 # row-major epilogue (LDS transpose, 16-byte stores): same arithmetic, same bits
                 "64x64x32/2x2/m32/dma/re", "64x64x16/2x2/m32/dma/re", "64x64x16/2x2/m32/dma3/re", "128x64x16/2x2/m32/dma/re",
                 "128x64x16/2x2/m32/dma3/re", "128x64x32/4x2/m32/dma/re", "128x128x16/4x2/m32/dma/re",
                 "128x128x32/4x2/m32/dma/re", "64x128x16/2x2/m32/dma/re", "64x128x32/2x2/m32/dma/re"):
        for rep in range(6):
            got = hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, cfg=names.index(name), split=1)
            assert torch.equal(got, base), "%s differs from the register-staged result (rep %d)" % (name, rep)
    # 16x16x4 family among themselves
    base16 = hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, cfg=names.index("112x64x32/1x4/m16"), split=1)
    for name in ("32x64x32/2x2/m16/dma", "112x64x32/1x4/m16/dma", "32x128x32/2x2/m16/dma", "64x32x32/2x2/m16/dma",
                 "32x64x32/2x2/m16/dma3", "32x64x32/2x2/m16/dma4", "32x128x32/2x2/m16/dma4",
                 "32x64x64/2x2/m16/dma", "32x128x64/2x2/m16/dma",
                 "32x64x32/2x2/m16/dma/re", "32x64x64/2x2/m16/dma/re", "32x128x32/2x2/m16/dma/re", "64x32x32/2x2/m16/dma/re",
                 "112x64x32/1x4/m16/dma/re", "64x144x32/4x1/m16/dma", "128x144x32/8x1/m16/dma"):
        for rep in range(6):
            got = hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, cfg=names.index(name), split=1)
            assert torch.equal(got, base16), "%s differs (rep %d)" % (name, rep)


def test_conv_ragged_channels_every_config(ptx):
    lib = _lib(ptx)
    x, w = rnd(2, 51, 3, 7, 6, seed=8), rnd(85, 51, 1, 3, 3, seed=9, scale=0.05)
    bias = rnd(85, seed=10)
    want = ref_conv(x, w, (1, 1, 1), (0, 1, 1), bias=bias)
    for cfg in fp32_configs(lib):
        close(hip_conv(ptx, x, w, (1, 1, 1), (0, 1, 1), bias=bias, cfg=cfg, split=1), want)


def test_conv_epilogues(ptx):
    x, w = rnd(2, 64, 4, 8, 8, seed=11), rnd(128, 64, 1, 1, 1, seed=12, scale=0.1)
    bn = make_bn(128, 13)
    bias = rnd(128, seed=14)
    one, zero = (1, 1, 1), (0, 0, 0)
    close(hip_conv(ptx, x, w, one, zero), ref_conv(x, w, one, zero))                                # plain
    close(hip_conv(ptx, x, w, one, zero, bias=bias, bn=bn), ref_conv(x, w, one, zero, bias=bias, bn=bn))  # conv bias + BN
    # shortcut A, stride 1 (channel zero-pad only) and stride 2 (subsample + zero-pad)
    rp = rnd(2, 64, 4, 8, 8, seed=15)
    close(hip_conv(ptx, x, w, one, zero, bn=bn, relu=True, res_pad=rp, res_stride=1),
          ref_conv(x, w, one, zero, bn=bn, relu=True, res_pad=rp, res_stride=1))
    w3 = rnd(128, 64, 3, 3, 3, seed=16, scale=0.03)
    rp2 = rnd(2, 48, 4, 8, 8, seed=17)
    close(hip_conv(ptx, x, w3, (2, 2, 2), one, bn=bn, relu=True, res_pad=rp2, res_stride=2, split=3),
          ref_conv(x, w3, (2, 2, 2), one, bn=bn, relu=True, res_pad=rp2, res_stride=2))


def test_dual_source_conv(ptx):
    """ptx_conv3d_dual_fwd == relu(bn3(conv3(o)) + bn_d(conv_d(x)[stride s])) (resnet3D.py:135-142, :176-185),
    i.e. a bottleneck's last conv with its shortcut-B branch folded in as extra K columns."""
    L, lib = ptx._lib, _lib(ptx)
    null = C.c_void_p(0)
    for (N, T, H, W, C1, C2, Co, s_) in [(2, 4, 8, 8, 64, 64, 256, 1), (2, 2, 7, 7, 128, 256, 512, 2), (1, 1, 5, 6, 20, 36, 72, 2)]:
        T2, H2, W2 = (T - 1) * s_ + 1, (H - 1) * s_ + 1 + (s_ - 1), (W - 1) * s_ + 1
        o, x = rnd(N, C1, T, H, W, seed=60), rnd(N, C2, T2, H2, W2, seed=61)
        w3, wd = rnd(Co, C1, 1, 1, 1, seed=62, scale=C1 ** -0.5), rnd(Co, C2, 1, 1, 1, seed=63, scale=C2 ** -0.5)
        bn3, bnd = make_bn(Co, 64), make_bn(Co, 65)
        want = ref_conv(o, w3, (1, 1, 1), (0, 0, 0), bn=bn3) + ref_conv(x, wd, (s_, s_, s_), (0, 0, 0), bn=bnd)
        want = F.relu(want)
        Kc, Kc2, Co_pad = _r4(C1), _r4(C2), (Co + 127) // 128 * 128
        ld = Kc + Kc2
        wp = torch.full((Co_pad * ld,), float("nan"), device=DEV)
        bp = torch.full((Co_pad,), float("nan"), device=DEV)
        for (wt, bn, Ci_, kc, koff, acc) in ((w3, bn3, C1, Kc, 0, 0), (wd, bnd, C2, Kc2, Kc, 1)):
            pd = L.PackDesc(Co, Ci_, 1, 1, 1, kc, Co_pad, 0, ld, koff, acc)
            ts = [t.to(DEV) for t in bn[:4]]
            wdv = wt.contiguous().to(DEV)
            L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wdv), null, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                             C.c_float(1e-5), _p(wp), _p(bp), _st()), "pack dual")
            torch.cuda.synchronize()
        od, xd = to_cl(o), to_cl(x)
        ldy = _r4(Co)
        for cfg, split in ((-1, 0), (2, 1), (24, 1), (28, 2), (30, 1), (44, 1)):
            yd = torch.full((N, T, H, W, ldy), float("nan"), device=DEV)
            d = L.ConvDesc()
            d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, C1, od.shape[-1]
            d.To, d.Ho, d.Wo, d.Co, d.ldy = T, H, W, Co, ldy
            d.kT = d.kH = d.kW = d.sT = d.sH = d.sW = 1
            d.Kc, d.Co_pad, d.flags = Kc, Co_pad, L.PTX_EPI_RELU
            d.x2_C, d.x2_ld, d.x2_T, d.x2_H, d.x2_W = C2, xd.shape[-1], T2, H2, W2
            d.x2_sT = d.x2_sH = d.x2_sW = s_
            ws_bytes = lib.ptx_conv3d_workspace_bytes(C.byref(d), 8)
            ws = torch.empty(max(ws_bytes // 4, 4), device=DEV)
            L.check(lib.ptx_conv3d_dual_fwd(C.byref(d), _p(od), _p(xd), _p(wp), _p(bp), _p(yd), _p(ws), ws_bytes, cfg,
                                            split, _st()), "dual conv")
            torch.cuda.synchronize()
            close(from_cl(yd, Co), want)
    # the second source may not be combined with a residual operand
    d.flags = L.PTX_EPI_RES_ADD
    assert lib.ptx_conv3d_dual_fwd(C.byref(d), _p(od), _p(xd), _p(wp), _p(bp), _p(yd), None, 0, -1, 1, _st()) == 1


def test_stem_fold_path(ptx):
    """ptx_fold_kw_ncdhw + fold_kw weight pack + (7,7,1) conv == Conv3d(3,64,7,s(1,2,2),p3) + BN + ReLU
    (resnet3D.py:153-155)."""
    L, lib = ptx._lib, _lib(ptx)
    N, T, H, W = 2, 5, 30, 26
    x, w = rnd(N, 3, T, H, W, seed=20), rnd(64, 3, 7, 7, 7, seed=21, scale=0.03)
    bn = make_bn(64, 22)
    want = ref_conv(x, w, (1, 2, 2), (3, 3, 3), bn=bn, relu=True)
    Wo = (W + 6 - 7) // 2 + 1
    Ho = (H + 6 - 7) // 2 + 1
    xd = x.to(DEV)
    x2 = torch.full((N, T, H, Wo, 24), float("nan"), device=DEV)
    L.check(lib.ptx_fold_kw_ncdhw(_p(xd), _p(x2), N, 3, T, H, W, 7, 2, 3, Wo, 24, _st()), "fold")
    pd = L.PackDesc(64, 3, 7, 7, 7, 24, 128, 1)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(128, device=DEV)
    ts = [t.to(DEV) for t in bn[:4]]
    wd = w.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), None, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                     C.c_float(1e-5), _p(wp), _p(bp), _st()), "pack")
    # live = 21 selects the 11-MFMA-per-tap stem path on BK = 24 tiles; 24 the generic one
    for cfg, live in ((-1, 21), (6, 21), (7, 24), (16, 21), (17, 21), (17, 24), (1, 21), (9, 24)):
        yd = torch.full((N, T, Ho, Wo, 64), float("nan"), device=DEV)
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, Wo, live, 24
        d.To, d.Ho, d.Wo, d.Co, d.ldy = T, Ho, Wo, 64, 64
        d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 7, 7, 1, 1, 2, 1, 3, 3, 0
        d.Kc, d.Co_pad, d.flags = 24, 128, L.PTX_EPI_RELU
        L.check(lib.ptx_conv3d_fwd(C.byref(d), _p(x2), _p(wp), _p(bp), None, _p(yd), None, 0, cfg, 1, _st()), "conv")
        torch.cuda.synchronize()
        close(from_cl(yd, 64), want)


def test_maxpool(ptx):
    L, lib = ptx._lib, _lib(ptx)
    for (N, T, H, W, Cc, k, s, p) in [(2, 6, 13, 12, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                      (1, 1, 15, 15, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
                                      (2, 5, 8, 8, 20, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                      (1, 3, 9, 21, 24, (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # stride-1 sliding windows
                                      (1, 2, 6, 19, 8, (1, 3, 3), (1, 1, 1), (0, 1, 0)),
                                      (1, 2, 5, 35, 12, (2, 3, 3), (2, 2, 2), (1, 0, 1)),
                                      (1, 4, 6, 19, 8, (2, 2, 3), (2, 2, 1), (1, 1, 1)),       # W-sliding kernel (kH != 3)
                                      (1, 3, 7, 21, 8, (1, 5, 3), (1, 1, 2), (0, 2, 1)),
                                      (1, 2, 6, 9, 8, (1, 3, 2), (1, 1, 2), (0, 1, 1))]:       # generic kernel (kW != 3)
        x = rnd(N, Cc, T, H, W, seed=30)
        want = F.max_pool3d(x, k, s, p)
        xd = to_cl(x)
        To, Ho, Wo = want.shape[2:]
        yd = torch.full((N, To, Ho, Wo, xd.shape[-1]), float("nan"), device=DEV)
        d = L.PoolDesc(N, T, H, W, Cc, xd.shape[-1], To, Ho, Wo, *k, *s, *p)
        L.check(lib.ptx_maxpool3d_fwd(C.byref(d), _p(xd), _p(yd), _st()), "maxpool")
        torch.cuda.synchronize()
        assert torch.equal(from_cl(yd, Cc), want)      # max is exact


def test_avgpool_and_linear(ptx):
    L, lib = ptx._lib, _lib(ptx)
    x = rnd(3, 100, 2, 7, 7, seed=31)
    want = F.adaptive_avg_pool3d(x, 1).flatten(1)
    xd = to_cl(x)
    out = torch.empty(3, 100, device=DEV)
    L.check(lib.ptx_global_avgpool(_p(xd), _p(out), 3, 100, 98, xd.shape[-1], 0, _st()), "avgpool cl")
    close(out.cpu(), want, 1e-5)
    xcf = x.to(DEV)
    out2 = torch.empty(3, 100, device=DEV)
    L.check(lib.ptx_global_avgpool(_p(xcf), _p(out2), 3, 100, 98, 100, 1, _st()), "avgpool cf")
    close(out2.cpu(), want, 1e-5)
    for (M, K, Nout, flags) in [(8, 2048, 339, 0), (1, 2048, 17, 0), (9, 100, 33, L.PTX_EPI_RELU),
                                (5, 62, 7, L.PTX_PRO_RELU | L.PTX_EPI_RELU), (3, 16384, 64, L.PTX_PRO_RELU)]:
        a, w, b = rnd(M, K, seed=32), rnd(Nout, K, seed=33, scale=K ** -0.5), rnd(Nout, seed=34)
        ain = F.relu(a) if flags & L.PTX_PRO_RELU else a
        want = F.linear(ain, w, b)
        want = F.relu(want) if flags & L.PTX_EPI_RELU else want
        ad, wd, bd = a.to(DEV), w.to(DEV), b.to(DEV)
        yd = torch.full((M, Nout), float("nan"), device=DEV)
        L.check(lib.ptx_linear_fwd(_p(ad), _p(wd), _p(bd), _p(yd), M, K, Nout, K, Nout, flags, _st()), "linear")
        close(yd.cpu(), want, 1e-5)
        L.check(lib.ptx_linear_fwd(_p(ad), _p(wd), None, _p(yd), M, K, Nout, K, Nout, flags | L.PTX_EPI_ACCUM, _st()),
                "linear accum")
        extra = F.linear(ain, w)
        # accumulate: y_new = act(y_old + xW) -- with RELU flags y_old >= 0 already
        want2 = want + extra
        want2 = F.relu(want2) if flags & L.PTX_EPI_RELU else want2
        close(yd.cpu(), want2, 1e-5)


def test_layout_transforms(ptx):
    L, lib = ptx._lib, _lib(ptx)
    for (N, Cc, S) in [(2, 3, 1000), (1, 70, 37), (3, 2048, 49)]:
        x = rnd(N, Cc, S, seed=40)
        ld = _r4(Cc)
        xd = x.to(DEV)
        y = torch.full((N, S, ld), float("nan"), device=DEV)
        L.check(lib.ptx_ncdhw_to_ndhwc(_p(xd), _p(y), N, Cc, S, ld, _st()), "to cl")
        torch.cuda.synchronize()
        assert torch.equal(y[..., :Cc].cpu(), x.permute(0, 2, 1))
        assert bool((y[..., Cc:] == 0).all())
        z = torch.full((N, Cc, S), float("nan"), device=DEV)
        L.check(lib.ptx_ndhwc_to_ncdhw(_p(y), _p(z), N, Cc, S, ld, _st()), "to cf")
        torch.cuda.synchronize()
        assert torch.equal(z.cpu(), x)
    x = rnd(2, 50, 24, seed=41)      # [batch][R=50][ldx=24], Cc=22 valid
    xd = x.to(DEV)
    y = torch.full((2, 22, 52), float("nan"), device=DEV)
    L.check(lib.ptx_transpose_last2(_p(xd), _p(y), 2, 50, 22, 24, 52, _st()), "transpose")
    torch.cuda.synchronize()
    assert torch.equal(y[:, :, :50].cpu(), x[:, :, :22].permute(0, 2, 1))
    assert bool((y[:, :, 50:] == 0).all())


def test_softmax_and_bgemm(ptx):
    L, lib = ptx._lib, _lib(ptx)
    for cols in (196, 1568, 37):
        ld = _r4(cols)
        x = rnd(5, ld, seed=42, scale=3.0)
        xd = x.clone().to(DEV)
        L.check(lib.ptx_softmax_rows(_p(xd), 5, cols, ld, 0, _st()), "softmax")
        torch.cuda.synchronize()
        close(xd[:, :cols].cpu(), F.softmax(x[:, :cols], -1), 1e-6)
        assert bool((xd[:, cols:] == 0).all())
        xs = x.clone().to(DEV)
        L.check(lib.ptx_softmax_rows(_p(xs), 5, cols, ld, 1, _st()), "scale")
        close(xs[:, :cols].cpu(), x[:, :cols] / cols, 1e-6)
    for (B, M, Nn, K) in [(3, 196, 196, 64), (2, 200, 100, 52), (2, 300, 256, 1568), (1, 70, 33, 10)]:
        lda, ldb, ldc = _r4(K) + 4, _r4(K), _r4(Nn)
        a = torch.zeros(B, M, lda)
        b = torch.zeros(B, Nn, ldb)
        a[..., :K] = rnd(B, M, K, seed=43)
        b[..., :K] = rnd(B, Nn, K, seed=44)
        ad, bd = a.to(DEV), b.to(DEV)
        cd = torch.full((B, M, ldc), float("nan"), device=DEV)
        L.check(lib.ptx_bgemm_nt(_p(ad), _p(bd), _p(cd), B, M, Nn, K, lda, ldb, ldc, M * lda, Nn * ldb, M * ldc, _st()),
                "bgemm")
        torch.cuda.synchronize()
        close(cd[..., :Nn].cpu(), torch.matmul(a[..., :K], b[..., :K].transpose(1, 2)), 1e-4)
        assert bool((cd[..., Nn:] == 0).all())


def test_error_paths_on_device(ptx):
    L, lib = ptx._lib, _lib(ptx)
    x = torch.zeros(128 * 64, device=DEV)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = 1, 1, 4, 4, 64, 64
    d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, 4, 4, 4, 4
    d.kT = d.kH = d.kW = 1
    d.sT = d.sH = d.sW = 1
    d.Kc, d.Co_pad = 64, 128
    assert lib.ptx_conv3d_fwd(C.byref(d), _p(x), _p(x), None, None, _p(x, 4096), None, 0, 2, 1, _st()) == 0
    # split-K without workspace must be refused, not crash
    st = lib.ptx_conv3d_fwd(C.byref(d), _p(x), _p(x), None, None, _p(x, 4096), None, 0, 2, 2, _st())
    assert st == 4 and b"workspace" in lib.ptx_last_error()
    d.flags = L.PTX_EPI_RES_ADD
    st = lib.ptx_conv3d_fwd(C.byref(d), _p(x), _p(x), None, None, _p(x, 4096), None, 0, 2, 1, _st())
    assert st == 1 and b"residual" in lib.ptx_last_error()
    with pytest.raises(L.PtxError):
        L.check(st, "conv")
    # misaligned pointer
    d.flags = 0
    st = lib.ptx_conv3d_fwd(C.byref(d), _p(x, 1), _p(x), None, None, _p(x, 4096), None, 0, 2, 1, _st())
    assert st == 1 and b"aligned" in lib.ptx_last_error()
    torch.cuda.synchronize()


def test_conv_output_channel_slice(ptx):
    """A conv writing a channel slice of a wider tensor (row stride ldy > Co): the in-place form of
    torch.cat(dim=1) (slowfast.py:145-151) -- neighbours untouched, also through split-K."""
    L, lib = ptx._lib, _lib(ptx)
    N, T, H, W, Ci, Co, c0, total = 2, 3, 9, 10, 32, 16, 64, 96
    x, w = rnd(N, Ci, T, H, W, seed=70), rnd(Co, Ci, 5, 1, 1, seed=71, scale=0.1)
    want = F.conv3d(x, w, None, (2, 1, 1), (2, 0, 0))
    To = want.shape[2]
    pd = L.PackDesc(Co, Ci, 5, 1, 1, Ci, 128, 0)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(128, device=DEV)
    wd = w.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), None, None, None, None, None, C.c_float(0), _p(wp), _p(bp),
                                     _st()), "pack")
    xd = to_cl(x)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, Ci
    d.To, d.Ho, d.Wo, d.Co, d.ldy = To, H, W, Co, total
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 5, 1, 1, 2, 1, 1, 2, 0, 0
    d.Kc, d.Co_pad = Ci, 128
    ws_bytes = lib.ptx_conv3d_workspace_bytes(C.byref(d), 4)
    assert ws_bytes == 4 * N * To * H * W * Co * 4            # dense [M][Co] slabs, not [M][ldy]
    ws = torch.empty(ws_bytes // 4, device=DEV)
    for cfg, split in ((-1, 0), (28, 1), (30, 4), (0, 2)):
        cat = torch.full((N, To, H, W, total), 7.0, device=DEV)
        L.check(lib.ptx_conv3d_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), None, _p(cat, c0), _p(ws), ws_bytes, cfg, split,
                                   _st()), "conv slice")
        torch.cuda.synchronize()
        got = cat.cpu()
        close(got[..., c0:c0 + Co].permute(0, 4, 1, 2, 3), want)
        assert bool((got[..., :c0] == 7.0).all()) and bool((got[..., c0 + Co:] == 7.0).all()), (cfg, split)


def test_fold_strided_and_frames_u8(ptx):
    """ptx_fold_kw_strided on `input[:, :, ::step]` == fold of the materialised slice (bit-equal);
    ptx_fold_kw_frames_u8 / ptx_frames_u8_to_ncdhw == TransformImage's tensor half (utils.py:72-75)
    followed by the fold, bit-equal to the CPU fp32 arithmetic."""
    from oracle import functional as OF
    L, lib = ptx._lib, _lib(ptx)
    N, T, H, W, step = 2, 11, 13, 18, 4
    Wo = (W + 6 - 7) // 2 + 1
    x = rnd(N, 3, T, H, W, seed=80)
    xs = x[:, :, ::step].contiguous()
    Ts = xs.shape[2]
    a = torch.full((N, Ts, H, Wo, 24), float("nan"), device=DEV)
    b = torch.full((N, Ts, H, Wo, 24), float("nan"), device=DEV)
    xd, xsd = x.to(DEV), xs.to(DEV)
    L.check(lib.ptx_fold_kw_ncdhw(_p(xsd), _p(a), N, 3, Ts, H, W, 7, 2, 3, Wo, 24, _st()), "fold")
    L.check(lib.ptx_fold_kw_strided(_p(xd), _p(b), N, 3, Ts, H, W, 3 * T * H * W, T * H * W, step * H * W, 7, 2, 3, Wo,
                                    24, _st()), "fold strided")
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    # reference of the fold itself: y[n,t,h,wo,kw*3+c] = x[n,c,t,h,wo*2-3+kw]
    xp = F.pad(xs, (3, 3))
    ref = torch.zeros(N, Ts, H, Wo, 24)
    for kw in range(7):
        ref[..., kw * 3:kw * 3 + 3] = xp[:, :, :, :, kw:kw + 2 * Wo:2][..., :Wo].permute(0, 2, 3, 4, 1)
    assert torch.equal(a.cpu(), ref)
    g = torch.Generator().manual_seed(81)
    frames = torch.randint(0, 256, (N, T, H, W, 3), dtype=torch.uint8, generator=g)
    fd = frames.to(DEV)
    for space, rng_ in (("RGB", [0, 1]), ("BGR", [0, 255])):
        mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
        if rng_[1] == 255:
            mean, std = [104.0, 117.0, 123.0], [1.0, 57.0, 58.5]
        nd = L.NormDesc.make(mean, std, space, rng_)
        want = OF.transform_frames(frames, mean, std, space, rng_)
        y = torch.full((N, 3, T, H, W), float("nan"), device=DEV)
        L.check(lib.ptx_frames_u8_to_ncdhw(C.c_void_p(fd.data_ptr()), _p(y), N, T, H, W, 3, C.byref(nd), _st()), "u8")
        torch.cuda.synchronize()
        assert torch.equal(y.cpu(), want), space                 # same fp32 ops in the same order
        got = ptx.transforms.FramesToTensor(dict(mean=mean, std=std, input_space=space, input_range=rng_))(fd)
        assert torch.equal(got.cpu(), want)
        wd = want[:, :, ::step].contiguous().to(DEV)
        L.check(lib.ptx_fold_kw_ncdhw(_p(wd), _p(a), N, 3, Ts, H, W, 7, 2, 3, Wo, 24, _st()), "fold")
        L.check(lib.ptx_fold_kw_frames_u8(C.c_void_p(fd.data_ptr()), _p(b), N, 3, Ts, H, W, step, T, 7, 2, 3, Wo, 24,
                                          C.byref(nd), _st()), "fold u8")
        torch.cuda.synchronize()
        assert torch.equal(a, b), space


def test_maxpool_same_slices_copy_and_window_mean(ptx):
    L, lib = ptx._lib, _lib(ptx)
    # TF-"SAME" pooling with zero-valued padding (F.pad + MaxPool3d), output into a channel slice
    for (N, T, H, W, Cc, k, s) in [(2, 5, 14, 14, 24, (1, 3, 3), (1, 2, 2)), (1, 7, 9, 11, 16, (3, 3, 3), (2, 2, 2)),
                                   (2, 4, 7, 7, 32, (2, 2, 2), (2, 2, 2)), (1, 4, 6, 6, 12, (3, 3, 3), (1, 1, 1)),
                                   (1, 3, 10, 28, 16, (3, 3, 3), (1, 1, 1)), (1, 4, 9, 37, 8, (3, 3, 3), (2, 2, 2)),
                                   (4, 16, 26, 27, 512, (3, 3, 3), (1, 1, 1))]:     # enough threads for the 4 x 4 patches
        x = rnd(N, Cc, T, H, W, seed=90) - 0.5
        out = [-(-i // st) for i, st in zip((T, H, W), s)]
        tot = [max((o - 1) * st + kk - i, 0) for o, st, kk, i in zip(out, s, k, (T, H, W))]
        fr = [t // 2 for t in tot]
        xp = F.pad(x, (fr[2], tot[2] - fr[2], fr[1], tot[1] - fr[1], fr[0], tot[0] - fr[0]))
        want = F.max_pool3d(xp, k, s)
        assert list(want.shape[2:]) == out
        xd = to_cl(x)
        total, c0 = Cc + 40, 8
        cat = torch.full((N, *out, total), 3.0, device=DEV)
        d = L.PoolDesc(N, T, H, W, Cc, xd.shape[-1], *out, *k, *s, *fr, total, L.PTX_POOL_SAME | L.PTX_POOL_PAD_ZERO)
        L.check(lib.ptx_maxpool3d_fwd(C.byref(d), _p(xd), _p(cat, c0), _st()), "maxpool same")
        torch.cuda.synchronize()
        got = cat.cpu()
        assert torch.equal(got[..., c0:c0 + Cc].permute(0, 4, 1, 2, 3), want)
        assert bool((got[..., :c0] == 3.0).all()) and bool((got[..., c0 + Cc:] == 3.0).all())
    # regular pooling into a slice through the sliding-window kernel (slowfast.py:123 + :145)
    x = rnd(2, 64, 2, 20, 20, seed=91)
    want = F.max_pool3d(x, (1, 3, 3), (1, 2, 2), (0, 1, 1))
    xd = to_cl(x)
    cat = torch.full((2, 2, 10, 10, 80), 3.0, device=DEV)
    d = L.PoolDesc(2, 2, 20, 20, 64, 64, 2, 10, 10, 1, 3, 3, 1, 2, 2, 0, 1, 1, 80, 0)
    L.check(lib.ptx_maxpool3d_fwd(C.byref(d), _p(xd), _p(cat), _st()), "maxpool slice")
    torch.cuda.synchronize()
    assert torch.equal(cat.cpu()[..., :64].permute(0, 4, 1, 2, 3), want) and bool((cat[..., 64:] == 3.0).all())
    # copy2d: place [rows, cols] into a column window
    src = rnd(7, 24, seed=92).to(DEV)
    dst = torch.zeros(7, 40, device=DEV)
    L.check(lib.ptx_copy2d(_p(src), _p(dst, 12), 7, 24, 24, 40, _st()), "copy2d")
    torch.cuda.synchronize()
    assert torch.equal(dst[:, 12:36], src) and bool((dst[:, :12] == 0).all()) and bool((dst[:, 36:] == 0).all())
    # window mean: avg_pool over time (k=2, s=1) and mean over all steps
    x = rnd(3, 8, 20, seed=93)
    xd = x.to(DEV)
    y = torch.empty(3, 7, 20, device=DEV)
    L.check(lib.ptx_window_mean(_p(xd), _p(y), 3, 8, 20, 2, 1, _st()), "window_mean")
    z = torch.empty(3, 1, 20, device=DEV)
    L.check(lib.ptx_window_mean(_p(xd), _p(z), 3, 8, 20, 8, 1, _st()), "window_mean all")
    torch.cuda.synchronize()
    close(y.cpu(), (x[:, :-1] + x[:, 1:]) / 2, tol=1e-6)
    close(z.cpu()[:, 0], x.mean(1), tol=1e-6)
    assert lib.ptx_window_mean(_p(xd), _p(z), 3, 8, 20, 9, 1, _st()) == 1
    assert lib.ptx_copy2d(_p(src), _p(dst), 7, 22, 24, 40, _st()) == 1


def test_skinny_linear_gather_and_setsum(ptx):
    """The weight-bandwidth-bound small-M GEMM (classifier heads, TRN MLPs): dense, in-kernel frame
    gather (trn.py:104-108) and the set-summed second Linear (trn.py:110 by linearity)."""
    L, lib = ptx._lib, _lib(ptx)
    for (M, K, N, flags) in [(8, 2048, 339, 0), (3, 512, 10, L.PTX_EPI_RELU), (24, 1024, 64, L.PTX_PRO_RELU),
                             (1, 16384, 1024, L.PTX_PRO_RELU | L.PTX_EPI_RELU), (13, 36, 7, 0), (5, 30, 9, 0)]:
        x, w, b = rnd(M, K, seed=100), rnd(N, K, seed=101, scale=K ** -0.5), rnd(N, seed=102)
        xi = F.relu(x) if flags & L.PTX_PRO_RELU else x
        want = xi @ w.t() + b
        want = F.relu(want) if flags & L.PTX_EPI_RELU else want
        xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
        y = torch.full((M, N + 3), 5.0, device=DEV)
        L.check(lib.ptx_linear_fwd(_p(xd), _p(wd), _p(bd), _p(y), M, K, N, K, N + 3, flags, _st()), "linear")
        L.check(lib.ptx_linear_fwd(_p(xd), _p(wd), None, _p(y), M, K, N, K, N + 3, (flags & L.PTX_PRO_RELU) | L.PTX_EPI_ACCUM,
                                   _st()), "linear accum")
        torch.cuda.synchronize()
        close(y[:, :N].cpu(), want + xi @ w.t(), tol=2e-5)
        assert bool((y[:, N:] == 5.0).all())
    # frame gather: 3 subsets of 4 frames out of 8, B = 5 videos
    B, T, Fd, hid = 5, 8, 64, 48
    x, w, b = rnd(B, T, Fd, seed=103), rnd(hid, 4 * Fd, seed=104, scale=0.06), rnd(hid, seed=105)
    subsets = [(0, 2, 3, 7), (1, 2, 5, 6), (0, 1, 2, 3)]
    d = L.RelationDesc()
    d.B, d.n_sets, d.n_frames, d.frame_len = B, 3, 4, Fd
    for r, sub in enumerate(subsets):
        for f, i in enumerate(sub):
            d.idx[r][f] = i
    want = torch.cat([F.relu(F.relu(x[:, list(sub)].reshape(B, -1)) @ w.t() + b) for sub in subsets], 0)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    h = torch.empty(3 * B, hid, device=DEV)
    L.check(lib.ptx_relation_linear_fwd(C.byref(d), _p(xd), T * Fd, _p(wd), _p(bd), _p(h), hid, hid,
                                        L.PTX_PRO_RELU | L.PTX_EPI_RELU, _st()), "relation linear")
    torch.cuda.synchronize()
    close(h.cpu(), want, tol=2e-5)
    # second Linear over the summed subsets == sum of per-subset Linears
    w2, b2 = rnd(20, hid, seed=106, scale=0.1), rnd(20, seed=107)
    want2 = sum(want[r * B:(r + 1) * B] @ w2.t() + b2 for r in range(3))
    w2d, b2d = w2.to(DEV), b2.to(DEV)
    out = torch.ones(B, 20, device=DEV)
    L.check(lib.ptx_linear_setsum_fwd(_p(h), _p(w2d), _p(b2d), _p(out), B, 3, hid, 20, hid, 20, L.PTX_EPI_ACCUM, _st()),
            "setsum")
    torch.cuda.synchronize()
    close(out.cpu(), want2 + 1.0, tol=2e-5)
    d.idx[2][3] = 8                                             # frame outside the video
    assert lib.ptx_relation_linear_fwd(C.byref(d), _p(xd), T * Fd, _p(wd), _p(bd), _p(h), hid, hid, 0, _st()) == 1
    d.n_sets = 9
    assert lib.ptx_relation_linear_fwd(C.byref(d), _p(xd), T * Fd, _p(wd), _p(bd), _p(h), hid, hid, 0, _st()) == 1


def test_cbn_fold_affine_upsample_and_upsampled_skip(ptx):
    """Generator-stage kernels: class-conditional BN folded to a per-sample affine, cBN+ReLU+nearest
    upsample in one pass, tanh, and the GBlock skip `upsample(x[:, :Cout])` as a conv-epilogue gather."""
    L, lib = ptx._lib, _lib(ptx)
    N, H, W, Cc, tot, off = 3, 5, 6, 10, 32, 12
    x = rnd(N, Cc, 1, H, W, seed=110)
    gain, bias = rnd(N, tot, seed=111, scale=0.3), rnd(N, tot, seed=112, scale=0.3)
    mean, var = rnd(tot, seed=113, scale=0.2), torch.rand(tot, generator=torch.Generator().manual_seed(114)) + 0.5
    gd, bd, md, vd = gain.to(DEV), bias.to(DEV), mean.to(DEV), var.to(DEV)
    sc, sh = torch.empty(N, tot, device=DEV), torch.empty(N, tot, device=DEV)
    L.check(lib.ptx_cbn_fold(_p(gd), _p(bd), _p(md), _p(vd), C.c_float(1e-5), _p(sc), _p(sh), N, tot, tot, tot, tot, 1, _st()), "fold")
    xd = to_cl(x)
    for up, act in ((1, 1), (2, 1), (2, 0), (1, 2)):
        y = torch.full((N, 1, H * up, W * up, xd.shape[-1]), float("nan"), device=DEV)
        L.check(lib.ptx_affine_act_upsample(_p(xd), _p(y), _p(sc, off), _p(sh, off), tot, N, H, W, Cc, xd.shape[-1],
                                            y.shape[-1], up, act, _st()), "affine")
        torch.cuda.synchronize()
        g_, b_ = gain[:, off:off + Cc], bias[:, off:off + Cc]
        ref = F.batch_norm(x[:, :, 0], mean[off:off + Cc], var[off:off + Cc], None, None, False, 0.1, 1e-5)
        ref = ref * (1 + g_)[:, :, None, None] + b_[:, :, None, None]
        ref = F.relu(ref) if act == 1 else (torch.tanh(ref) if act == 2 else ref)
        ref = F.interpolate(ref, scale_factor=up) if up > 1 else ref
        close(from_cl(y, Cc)[:, :, 0], ref, tol=1e-5)
        assert bool((y[..., Cc:] == 0).all())
    # plain BN (gamma/beta shared by all samples): gain row stride 0, no "+1"
    gam, bet = rnd(Cc, seed=115) + 2, rnd(Cc, seed=116)
    gmd, btd = gam.to(DEV), bet.to(DEV)
    s2, h2 = torch.empty(N, Cc, device=DEV), torch.empty(N, Cc, device=DEV)
    L.check(lib.ptx_cbn_fold(_p(gmd), _p(btd), _p(md), _p(vd), C.c_float(1e-5), _p(s2), _p(h2), N, Cc, 0, 0, Cc, 0, _st()), "bn fold")
    y = torch.empty(N, 1, H, W, xd.shape[-1], device=DEV)
    L.check(lib.ptx_affine_act_upsample(_p(xd), _p(y), _p(s2), _p(h2), Cc, N, H, W, Cc, xd.shape[-1], y.shape[-1], 1, 0, _st()), "bn")
    torch.cuda.synchronize()
    close(from_cl(y, Cc)[:, :, 0], F.batch_norm(x[:, :, 0], mean[:Cc], var[:Cc], gam, bet, False, 0.1, 1e-5), tol=1e-5)
    # 1x1 conv at 2x resolution + upsampled, channel-truncated skip
    Ci, Co, Cs = 8, 12, 20
    a, w, b = rnd(N, Ci, 1, 2 * H, 2 * W, seed=117), rnd(Co, Ci, 1, 1, 1, seed=118, scale=0.3), rnd(Co, seed=119)
    skip = rnd(N, Cs, 1, H, W, seed=120)
    want = F.conv3d(a, w, b) + F.interpolate(skip[:, :Co, 0], scale_factor=2)[:, :, None]
    pd = L.PackDesc(Co, Ci, 1, 1, 1, Ci, 128, 0)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(128, device=DEV)
    wd, bdv = w.to(DEV), b.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bdv), None, None, None, None, C.c_float(0), _p(wp), _p(bp), _st()), "pack")
    ad, sd_ = to_cl(a), to_cl(skip)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, 1, 2 * H, 2 * W, Ci, Ci
    d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, 2 * H, 2 * W, Co, Co
    d.kT = d.kH = d.kW = d.sT = d.sH = d.sW = 1
    d.Kc, d.Co_pad = Ci, 128
    d.flags = L.PTX_EPI_RES_PADA | L.PTX_EPI_RES_UP
    d.ldr, d.res_C, d.res_T, d.res_H, d.res_W, d.res_sT, d.res_sH, d.res_sW = Cs, Cs, 1, H, W, 0, 1, 1
    ws = torch.empty(4 * N * 4 * H * W * Co, device=DEV)
    for cfg, split in ((-1, 0), (28, 1), (30, 2)):
        yd = torch.full((N, 1, 2 * H, 2 * W, Co), float("nan"), device=DEV)
        L.check(lib.ptx_conv3d_fwd(C.byref(d), _p(ad), _p(wp), _p(bp), _p(sd_), _p(yd), _p(ws), ws.numel() * 4, cfg, split,
                                   _st()), "conv up-skip")
        torch.cuda.synchronize()
        close(from_cl(yd, Co), want)
    d.res_sH = 5
    assert lib.ptx_conv3d_fwd(C.byref(d), _p(ad), _p(wp), _p(bp), _p(sd_), _p(yd), None, 0, -1, 1, _st()) == 1


def test_direct_narrow_conv(ptx):
    """The VALU direct kernels (narrow outputs: SlowFast fast pathway / lateral convs) against ATen and,
    bit for bit in k-order terms, within fp32 reorder noise of the MFMA tiles."""
    lib = _lib(ptx)
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    direct = [i for i, n in enumerate(names) if n.endswith("/direct") and not n.split("/")[0].endswith("x24")]
    assert len(direct) >= 4
    cases = [  # N,T,H,W, Ci,Co, k, s, p
        (2, 6, 13, 11, 8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # fast res2 conv2
        (2, 6, 13, 11, 32, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0)),      # fast res2 conv1 (head_conv 3)
        (1, 17, 9, 10, 8, 16, (5, 1, 1), (8, 1, 1), (2, 0, 0)),      # lateral conv
        (2, 5, 12, 14, 16, 16, (1, 3, 3), (1, 2, 2), (0, 1, 1)),     # strided
        (1, 3, 7, 9, 12, 20, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # ragged widths, two channel tiles
    ]
    for (N, T, H, W, Ci, Co, k, s, p) in cases:
        x, w = rnd(N, Ci, T, H, W, seed=130), rnd(Co, Ci, *k, seed=131, scale=(Ci * k[0] * k[1] * k[2]) ** -0.5)
        bn = make_bn(Co, 132)
        res = None
        if s == (1, 1, 1):
            res = rnd(N, Co, T, H, W, seed=133)
        want = ref_conv(x, w, s, p, bn=bn, relu=True, res=res)
        for cfg in direct:
            close(hip_conv(ptx, x, w, s, p, bn=bn, relu=True, res=res, cfg=cfg, split=1), want)
    # kW-folded stem (3 -> 8, (5,7,7)): the x24 direct kernels on the folded operand
    L = ptx._lib
    N, T, H, W = 2, 6, 22, 26
    x, w = rnd(N, 3, T, H, W, seed=134), rnd(8, 3, 5, 7, 7, seed=135, scale=0.05)
    bn = make_bn(8, 136)
    want = ref_conv(x, w, (1, 2, 2), (2, 3, 3), bn=bn, relu=True)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    xd = x.to(DEV)
    x2 = torch.empty((N, T, H, Wo, 24), device=DEV)
    L.check(lib.ptx_fold_kw_ncdhw(_p(xd), _p(x2), N, 3, T, H, W, 7, 2, 3, Wo, 24, _st()), "fold")
    pd = L.PackDesc(8, 3, 5, 7, 7, 24, 128, 1)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(128, device=DEV)
    ts = [t.to(DEV) for t in bn[:4]]
    wd = w.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), None, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                     C.c_float(1e-5), _p(wp), _p(bp), _st()), "pack")
    for cfg in [i for i, n in enumerate(names) if n.endswith("x24/direct")] + [16]:
        yd = torch.full((N, T, Ho, Wo, 8), float("nan"), device=DEV)
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, Wo, 21, 24
        d.To, d.Ho, d.Wo, d.Co, d.ldy = T, Ho, Wo, 8, 8
        d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 5, 7, 1, 1, 2, 1, 2, 3, 0
        d.Kc, d.Co_pad, d.flags = 24, 128, L.PTX_EPI_RELU
        L.check(lib.ptx_conv3d_fwd(C.byref(d), _p(x2), _p(wp), _p(bp), None, _p(yd), None, 0, cfg, 1, _st()), "conv")
        torch.cuda.synchronize()
        close(from_cl(yd, 8), want)


def test_grouped_conv(ptx):
    """Grouped 3x3x3 convolution (ResNeXt3D, cardinality 32: resnext3D.py:85-92) on the direct tiles."""
    L, lib = ptx._lib, _lib(ptx)
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    for (N, T, H, W, Cin, Co, G, s) in [(2, 4, 9, 10, 128, 128, 32, (1, 1, 1)), (1, 5, 11, 8, 256, 256, 32, (2, 2, 2)),
                                       (1, 3, 6, 7, 64, 64, 4, (1, 1, 1)), (1, 2, 5, 5, 256, 512, 16, (1, 2, 2))]:
        cig = Cin // G
        x, w = rnd(N, Cin, T, H, W, seed=140), rnd(Co, cig, 3, 3, 3, seed=141, scale=(cig * 27) ** -0.5)
        bn = make_bn(Co, 142)
        want = F.relu(F.batch_norm(F.conv3d(x, w, None, s, 1, 1, G), bn[2], bn[3], bn[0], bn[1], False, 0.1, bn[4]))
        To, Ho, Wo = want.shape[2:]
        pd = L.PackDesc(Co, cig, 3, 3, 3, _r4(cig), (Co + 127) // 128 * 128, 0)
        wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
        bp = torch.empty(pd.Co_pad, device=DEV)
        wd = w.to(DEV)
        ts = [t.to(DEV) for t in bn[:4]]
        L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), None, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                         C.c_float(bn[4]), _p(wp), _p(bp), _st()), "pack")
        xd = to_cl(x)
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Cin, Cin
        d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, Co, Co
        d.kT = d.kH = d.kW = 3
        d.sT, d.sH, d.sW = s
        d.pT = d.pH = d.pW = 1
        d.Kc, d.Co_pad, d.flags, d.groups = pd.Kc, pd.Co_pad, L.PTX_EPI_RELU, G
        ran = 0
        for cfg in [-1] + [i for i, n in enumerate(names) if n.endswith("/direct") and "x24/" not in n]:
            yd = torch.full((N, To, Ho, Wo, Co), float("nan"), device=DEV)
            st = lib.ptx_conv3d_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), None, _p(yd), None, 0, cfg, 1, _st())
            if st == 2:                                   # tile wider than the group: reported, not computed
                assert cfg >= 0
                continue
            assert st == 0, lib.ptx_last_error()
            torch.cuda.synchronize()
            close(from_cl(yd, Co), want)
            ran += 1
        assert ran >= 2
        assert lib.ptx_conv3d_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), None, _p(yd), None, 0, 28, 1, _st()) == 2   # MFMA tile
    d.groups = 3
    assert lib.ptx_conv3d_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), None, _p(yd), None, 0, -1, 1, _st()) == 1
    # narrow groups packed as block-diagonal 32-wide super-groups -> MFMA tiles inside one super-group
    mfma = [i for i, n in enumerate(names) if not n.endswith(("/direct", "/f16", "/x3")) and n.split("x")[1] in ("16", "32")]
    assert len(mfma) >= 4
    for (N, T, H, W, Cc, G, s_) in [(2, 4, 9, 10, 128, 32, (1, 1, 1)), (1, 5, 11, 8, 256, 32, (2, 2, 2)), (2, 3, 6, 6, 64, 4, (1, 1, 1))]:
        gw = Cc // G
        sub = 32 // gw
        x, w = rnd(N, Cc, T, H, W, seed=143), rnd(Cc, gw, 3, 3, 3, seed=144, scale=(gw * 27) ** -0.5)
        bias = rnd(Cc, seed=145)
        want = F.conv3d(x, w, bias, s_, 1, 1, G)
        To, Ho, Wo = want.shape[2:]
        pd = L.PackDesc(Cc, 32, 3, 3, 3, 32, (Cc + 127) // 128 * 128, 0, 0, 0, 0, sub, 32)
        wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
        bp = torch.empty(pd.Co_pad, device=DEV)
        wd, bd = w.to(DEV), bias.to(DEV)
        L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bd), None, None, None, None, C.c_float(0), _p(wp), _p(bp),
                                         _st()), "pack super-groups")
        xd = to_cl(x)
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Cc, Cc
        d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, Cc, Cc
        d.kT = d.kH = d.kW = 3
        d.sT, d.sH, d.sW = s_
        d.pT = d.pH = d.pW = 1
        d.Kc, d.Co_pad, d.groups = 32, pd.Co_pad, G // sub
        ws = torch.empty(4 * N * To * Ho * Wo * Cc, device=DEV)
        for cfg, split in [(-1, 0)] + [(c, 1) for c in mfma] + [(37, 3), (65, 1)]:
            yd = torch.full((N, To, Ho, Wo, Cc), float("nan"), device=DEV)
            L.check(lib.ptx_conv3d_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), None, _p(yd), _p(ws), ws.numel() * 4, cfg, split,
                                       _st()), "grouped mfma cfg %d" % cfg)
            torch.cuda.synchronize()
            close(from_cl(yd, Cc), want)
        assert lib.ptx_conv3d_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), None, _p(yd), None, 0, 28, 1, _st()) == 2   # 64-wide tile


def test_outer_sum_relu(ptx):
    L, lib = ptx._lib, _lib(ptx)
    a, b = rnd(3, 10, seed=150), rnd(3, 7, seed=151)
    ad, bd = a.to(DEV), b.to(DEV)
    f = torch.full((3, 10, 8), float("nan"), device=DEV)
    L.check(lib.ptx_outer_sum_relu(_p(ad), _p(bd), _p(f), 3, 10, 7, 8, _st()), "outer_sum_relu")
    torch.cuda.synchronize()
    want = F.relu(a[:, :, None] + b[:, None, :]) / 7
    close(f.cpu()[..., :7], want, tol=1e-6)
    assert bool((f[..., 7:] == 0).all())


def test_conv_f16_operands(ptx):
    """fp16-operand implicit GEMM (v_mfma_f32_32x32x16_f16 / 16x16x32_f16, fp32 accumulate / bias / residual /
    output): against an fp32 ATen conv of the SAME half-rounded inputs, so only the summation order differs."""
    L, lib = ptx._lib, _lib(ptx)
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    f16cfgs = [i for i, n in enumerate(names) if n.endswith("/f16") and "/kwr/" not in n]
    assert len(f16cfgs) >= 6
    for (N, H, W, Ci, Co, k, pad, relu, with_res) in [(2, 12, 10, 64, 96, 3, 1, True, True), (3, 9, 9, 40, 24, 1, 0, False, False),
                                                      (1, 20, 20, 128, 3, 3, 1, False, False), (2, 6, 6, 256, 160, 1, 0, True, True)]:
        x = rnd(N, Ci, 1, H, W, seed=160).half().float()
        w = rnd(Co, Ci, 1, k, k, seed=161, scale=(Ci * k * k) ** -0.5).half().float()
        bias = rnd(Co, seed=162)
        res = rnd(N, Co, 1, H, W, seed=163) if with_res else None
        want = F.conv3d(x, w, bias, 1, (0, pad, pad))
        if res is not None:
            want = want + res
        want = F.relu(want) if relu else want
        ldh = (Ci + 7) // 8 * 8
        xh = torch.zeros(N, 1, H, W, ldh, dtype=torch.float16)
        xh[..., :Ci] = x.permute(0, 2, 3, 4, 1).half()
        xd = xh.to(DEV)
        pd = L.PackDesc(Co, Ci, 1, k, k, ldh, (Co + 127) // 128 * 128, 0, 0, 0, 0, 0, 0, 1)
        wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV, dtype=torch.float16)
        bp = torch.empty(pd.Co_pad, device=DEV)
        wd, bd = w.to(DEV), bias.to(DEV)
        L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bd), None, None, None, None, C.c_float(0),
                                         C.c_void_p(wp.data_ptr()), _p(bp), _st()), "pack f16")
        rd = to_cl(res) if res is not None else None
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, 1, H, W, Ci // 2, ldh // 2          # words = channel pairs
        d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, H, W, Co, _r4(Co)
        d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 1, k, k, 1, 1, 1, 0, pad, pad
        d.Kc, d.Co_pad, d.groups = ldh // 2, pd.Co_pad, 1
        d.flags = L.PTX_F16_OPERANDS | (L.PTX_EPI_RELU if relu else 0) | (L.PTX_EPI_RES_ADD if res is not None else 0)
        d.ldr = rd.shape[-1] if rd is not None else 0
        ws = torch.empty(4 * N * H * W * _r4(Co), device=DEV)
        for cfg, split in [(-1, 0)] + [(c, 1) for c in f16cfgs] + [(f16cfgs[2], 3)]:
            yd = torch.full((N, 1, H, W, _r4(Co)), float("nan"), device=DEV)
            L.check(lib.ptx_conv3d_fwd(C.byref(d), C.c_void_p(xd.data_ptr()), C.c_void_p(wp.data_ptr()), _p(bp),
                                       _p(rd) if rd is not None else None, _p(yd), _p(ws), ws.numel() * 4, cfg, split, _st()),
                    "conv f16 cfg %d" % cfg)
            torch.cuda.synchronize()
            close(from_cl(yd, Co), want, tol=1e-4)
        assert lib.ptx_conv3d_fwd(C.byref(d), C.c_void_p(xd.data_ptr()), C.c_void_p(wp.data_ptr()), _p(bp),
                                  _p(rd) if rd is not None else None, _p(yd), None, 0, 28, 1, _st()) == 2      # fp32 tile


@pytest.mark.parametrize("case", [
    # B, Nq, Nk, d, dv, mode, description
    (2, 100, 100, 16, 16, "softmax", "tiny channels on the <64,64> tile, ragged query / key tails"),
    (2, 196, 196, 512, 512, "softmax", "layer3 of config 3: d = 512, dv = 512 split over blockIdx.y"),
    (1, 300, 300, 256, 256, "softmax", "layer2 width (config 3: N = 1568)"),
    (2, 256, 64, 32, 128, "softmax", "BigGAN attention: pooled keys, d = ch/8, dv = ch/2"),
    (3, 130, 17, 64, 192, "softmax", "sub_sample-style Nk != Nq, dv not a multiple of 64"),
    (2, 90, 90, 40, 24, "scale", "dot_product mode (f / N, no softmax)"),
    (1, 64, 1568, 256, 256, "softmax", "many key tiles: online-softmax rescaling"),
    (2, 256, 64, 64, 256, "softmax/f16", "PTX_NL_F16: BigGAN-256 attention shape on 16x16x16 f16 MFMAs"),
    (2, 100, 50, 32, 40, "softmax/f16", "PTX_NL_F16: ragged tails"),
    (2, 196, 196, 512, 512, "softmax/x3", "PTX_NL_X3: layer3 of config 3, split operands"),
    (1, 300, 300, 256, 256, "softmax/x3", "PTX_NL_X3: layer2 width"),
    (2, 90, 90, 40, 24, "scale/x3", "PTX_NL_X3: dot_product mode"),
    (1, 64, 1568, 256, 256, "softmax/x3", "PTX_NL_X3: many key tiles"),
    (2, 530, 200, 128, 24, "scale", "key split (8 waves, two key groups merged through LDS): dot-product mode"),
    (1, 530, 333, 200, 200, "softmax", "key split: ragged d / dv / queries / keys, a last tile whose second key group is empty"),
    (3, 600, 1000, 256, 256, "softmax", "key split: layer2 widths, several clips"),
    (2, 520, 160, 96, 128, "softmax/x3", "key split, split operands"),
    (2, 196, 196, 1024, 512, "softmax", "gaussian mode at the reference's layer3 width: theta = x, d = C = 1024 (theta from global)"),
    (1, 392, 49, 1024, 512, "softmax", "d = 1024 with sub-sampled keys"),
    (2, 100, 60, 640, 320, "scale", "512 < d < 1024, dot-product scaling"),
    (2, 90, 70, 4, 24, "relu", "concatenation mode: relu(a_i + b_j) / N as a 2-term dot product on 4-float rows"),
    (1, 200, 200, 4, 256, "relu/x3", "concatenation mode, split operands"),
    (2, 196, 196, 1024, 512, "softmax/x3", "d > 512 under an x3 plan: the exact fp32 kernel stands in"),
    # round 4: Nq <= 512 with 64 < d <= 512 runs the d-split kernel (16-query workgroups, channel axes spread over the waves);
    # the cases above with such extents exercise it too (config 3's layer3: 196 x 196 x 512 x 512)
    (1, 576, 1568, 256, 256, "softmax", "the 64-query kernel on many key tiles (Nq > 512 keeps it off the d-split kernel)"),
    (2, 50, 25, 200, 136, "softmax", "d-split <256,256,4>: ragged d / dv / queries / keys"),
    (2, 196, 98, 512, 512, "scale", "d-split <512,512,8>: dot-product mode, sub-sampled keys"),
    (1, 392, 392, 512, 512, "softmax", "d-split: layer3 of NonLocalResNet3D-50 at the reference's 16 x 224 x 224 input"),
    (3, 17, 300, 320, 400, "softmax", "d-split <512,512,8> with padded slices, one ragged query group per clip"),
    (2, 100, 60, 72, 264, "relu", "d-split, relu(S) / N weights"),
])
def test_fused_nonlocal_attention(ptx, case):
    """ptx_nonlocal_fwd against the reference's op sequence (nonlocalnet.py:143-166 / :192-211): matmul ->
    softmax (or / N) -> matmul in torch fp32 on the CPU.  Operands are channel SLICES of one fused projection
    tensor (row stride > d), as the engine passes them.  Tolerance: fp32 summation order + v_exp_f32."""
    L, lib = ptx._lib, _lib(ptx)
    B, Nq, Nk, d, dv, mode, _ = case
    g_ = torch.Generator().manual_seed(1000 + Nq + d)
    ld = _r4(2 * d + dv) + 4
    tpg_q = torch.randn(B, Nq, ld, generator=g_)
    tpg_k = torch.randn(B, Nk, ld, generator=g_)
    scale = 3.0 / d ** 0.5                   # logits of a few units: a peaky but non-degenerate softmax
    theta, phi, gv = tpg_q[..., :d] * scale, tpg_k[..., d:2 * d].clone(), tpg_k[..., 2 * d:2 * d + dv].clone()
    tpg_q[..., :d] = theta
    f = torch.matmul(theta, phi.transpose(1, 2))
    half, x3 = mode.endswith("/f16"), mode.endswith("/x3")
    mode = mode.split("/")[0]
    f = F.softmax(f, dim=-1) if mode == "softmax" else (F.relu(f) if mode == "relu" else f) / f.size(-1)
    want = torch.matmul(f, gv)
    tq, tk = tpg_q.to(DEV), tpg_k.to(DEV)
    ldy = _r4(dv) + 8
    y = torch.full((B, Nq, ldy), float("nan"), device=DEV)
    desc = L.NonlocalDesc()
    desc.batch, desc.Nq, desc.Nk, desc.d, desc.dv = B, Nq, Nk, d, dv
    desc.ld_theta = desc.ld_phi = desc.ld_g = ld
    desc.ld_y = ldy
    desc.bs_theta, desc.bs_phi, desc.bs_g, desc.bs_y = Nq * ld, Nk * ld, Nk * ld, Nq * ldy
    desc.mode = ((L.PTX_NL_SOFTMAX if mode == "softmax" else L.PTX_NL_SCALE) | (L.PTX_NL_RELU if mode == "relu" else 0) |
                 (L.PTX_NL_F16 if half else 0) | (L.PTX_NL_X3 if x3 else 0))
    assert lib.ptx_nonlocal_supported(C.byref(desc))
    L.check(lib.ptx_nonlocal_fwd(C.byref(desc), _p(tq), _p(tk, d), _p(tk, 2 * d), _p(y), _st()), "nonlocal")
    torch.cuda.synchronize()
    got = y.cpu()
    assert torch.isnan(got[..., dv:]).all()                 # columns beyond dv are left untouched
    err = (got[..., :dv] - want).abs().max().item()
    # fp16 operands: theta / phi / g / P rounded to 11 bits, fp32 accumulate (bound chosen by the builder)
    assert err <= (5e-3 if half else 2e-5) * max(1.0, want.abs().max().item()), (case, err)
    # unsupported widths are refused, not mis-computed
    desc.d = 2048
    assert not lib.ptx_nonlocal_supported(C.byref(desc))
    assert lib.ptx_nonlocal_fwd(C.byref(desc), _p(tq), _p(tk, d), _p(tk, 2 * d), _p(y), _st()) != 0


@pytest.mark.parametrize("case", [
    (2, 1568, 1568, 256, 256, "", "layer2 of config 3: 25 query tiles -> 32 chunks of 38.3 units per clip"),
    (1, 3136, 3136, 128, 128, "", "the reference's usual N at 16 x 224 x 224: 49 query tiles -> 64 chunks; <256,128> tiles"),
    (3, 530, 1000, 200, 136, "", "ragged everything: 9 query tiles (the last of 18 queries), 32 key tiles (the last of 8 keys)"),
    (2, 1568, 784, 256, 256, "x3", "split operands, sub-sampled keys"),
    (1, 600, 520, 96, 256, "x3", "split operands, ragged"),
])
def test_nonlocal_attention_stream_k(ptx, case, monkeypatch):
    """(PTX_NL_STREAMK=1: the form is off by default -- measured slower than the plain kernel)
    ptx_nonlocal_ws_fwd with the workspace ptx_nonlocal_workspace_bytes asks for: the stream-K form (the units of one clip
    in 32 / 64 equal chunks, partial (O, max, sum) blocks folded by a combine launch) against the op sequence in torch fp32,
    and against the plain kernel (same arithmetic, another summation order); per-clip results do not depend on the batch;
    without (or with too small) a workspace the call runs exactly what ptx_nonlocal_fwd runs."""
    L, lib = ptx._lib, _lib(ptx)
    B, Nq, Nk, d, dv, flavour, _ = case
    monkeypatch.setenv("PTX_NL_STREAMK", "1")
    g_ = torch.Generator().manual_seed(2000 + Nq + d)
    ld = _r4(2 * d + dv) + 4
    tpg_q = torch.randn(B, Nq, ld, generator=g_)
    tpg_k = torch.randn(B, Nk, ld, generator=g_)
    tpg_q[..., :d] *= 3.0 / d ** 0.5
    theta, phi, gv = tpg_q[..., :d], tpg_k[..., d:2 * d], tpg_k[..., 2 * d:2 * d + dv]
    want = torch.matmul(F.softmax(torch.matmul(theta, phi.transpose(1, 2)), dim=-1), gv)
    tq, tk = tpg_q.to(DEV), tpg_k.to(DEV)
    ldy = _r4(dv) + 8

    def desc_for(batch):
        desc = L.NonlocalDesc()
        desc.batch, desc.Nq, desc.Nk, desc.d, desc.dv = batch, Nq, Nk, d, dv
        desc.ld_theta = desc.ld_phi = desc.ld_g = ld
        desc.ld_y = ldy
        desc.bs_theta, desc.bs_phi, desc.bs_g, desc.bs_y = Nq * ld, Nk * ld, Nk * ld, Nq * ldy
        desc.mode = L.PTX_NL_SOFTMAX | (L.PTX_NL_X3 if flavour == "x3" else 0)
        return desc
    desc = desc_for(B)
    need = lib.ptx_nonlocal_workspace_bytes(C.byref(desc))
    chunks = 32 if (Nq + 63) // 64 <= 32 else 64
    assert need == B * chunks * 2 * (64 * (128 if dv <= 128 else 256) + 128) * 4
    ws = torch.full((need // 4 + 4,), float("nan"), device=DEV)

    def run(desc, wsp, nbytes, tq_=tq, tk_=tk):
        y = torch.full((desc.batch, Nq, ldy), float("nan"), device=DEV)
        L.check(lib.ptx_nonlocal_ws_fwd(C.byref(desc), _p(tq_), _p(tk_, d), _p(tk_, 2 * d), _p(y), wsp, nbytes, _st()), "nonlocal ws")
        torch.cuda.synchronize()
        return y.cpu()
    got = run(desc, _p(ws), need)
    assert torch.isnan(got[..., dv:]).all() and torch.isnan(ws[need // 4:]).all()
    bound = 2e-5 * max(1.0, want.abs().max().item())
    assert (got[..., :dv] - want).abs().max().item() <= bound, case
    plain = torch.full((B, Nq, ldy), float("nan"), device=DEV)
    L.check(lib.ptx_nonlocal_fwd(C.byref(desc), _p(tq), _p(tk, d), _p(tk, 2 * d), _p(plain), _st()), "nonlocal")
    torch.cuda.synchronize()
    assert (got[..., :dv] - plain.cpu()[..., :dv]).abs().max().item() <= bound
    # no workspace / a short one / a misaligned one: the plain kernels, bit for bit
    for wsp, nb in ((None, 0), (_p(ws), need - 16), (C.c_void_p(ws.data_ptr() + 4), need)):
        assert torch.equal(run(desc, wsp, nb)[..., :dv], plain.cpu()[..., :dv])
    # repeatable, and a clip's bits do not depend on the batch it arrives in (the split is per clip)
    assert torch.equal(run(desc, _p(ws), need)[..., :dv], got[..., :dv])
    if B > 1:
        d1 = desc_for(1)
        for b in range(B):
            one = run(d1, _p(ws), need, tq[b:b + 1].contiguous(), tk[b:b + 1].contiguous())
            assert torch.equal(one[0, :, :dv], got[b, :, :dv]), (case, b)
    # shapes the stream-K form does not cover ask for no workspace
    short = desc_for(B)
    short.Nq = 256
    short.bs_theta, short.bs_y = 256 * ld, 256 * ldy
    assert lib.ptx_nonlocal_workspace_bytes(C.byref(short)) == 0
    sc = desc_for(B)
    sc.mode = L.PTX_NL_SCALE
    assert lib.ptx_nonlocal_workspace_bytes(C.byref(sc)) == 0
    monkeypatch.delenv("PTX_NL_STREAMK")          # the default: no workspace asked for, the plain kernels whatever is passed
    assert lib.ptx_nonlocal_workspace_bytes(C.byref(desc)) == 0
    assert torch.equal(run(desc, _p(ws), need)[..., :dv], plain.cpu()[..., :dv])


def _fused_stage_case(ptx, N, H, W, Ci, Co, k, up2, affine, relu, tanh, out16, dual, skip, cfgs):
    """One ptx_conv3d_fused_fwd launch per tile configuration against the op sequence it replaces, in torch fp32 on
    the CPU with the SAME half-rounded operands: [nearest x2] -> conv -> (+ skip) -> [raw] -> affine -> relu | tanh."""
    L, lib = ptx._lib, _lib(ptx)
    pad = k // 2
    x = rnd(N, Ci, 1, H, W, seed=300 + Ci).half().float()
    w = rnd(Co, Ci, 1, k, k, seed=301, scale=(Ci * k * k) ** -0.5).half().float()
    bias = rnd(Co, seed=302)
    Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
    xin = F.interpolate(x[:, :, 0], scale_factor=2, mode="nearest").unsqueeze(2) if up2 else x
    v = F.conv3d(xin, w, bias, 1, (0, pad, pad))
    res_t, res_flags, res_ld = None, 0, 0
    d = L.ConvDesc()
    if skip is not None:
        kind, r16 = skip                                    # ("same" | "up", halfs?)
        if kind == "same":
            r = rnd(N, Co, 1, Ho, Wo, seed=303)
            r = r.half().float() if r16 else r
            v = v + r
            res_flags = L.PTX_EPI_RES_ADD
        else:                                               # GBlock skip: up(x_prev[:, :Co]) with more channels stored
            Cr = Co + 8
            r = rnd(N, Cr, 1, Ho // 2, Wo // 2, seed=304)
            r = r.half().float() if r16 else r
            v = v + F.interpolate(r[:, :Co, 0], scale_factor=2, mode="nearest").unsqueeze(2)
            res_flags = L.PTX_EPI_RES_PADA | L.PTX_EPI_RES_UP
            d.res_C, d.res_T, d.res_H, d.res_W = Cr, 1, Ho // 2, Wo // 2
            d.res_sT, d.res_sH, d.res_sW = 0, 1, 1
        C_r = r.shape[1]
        res_ld = (C_r + 7) // 8 * 8 if r16 else _r4(C_r)
        rt = torch.zeros(N, 1, r.shape[3], r.shape[4], res_ld, dtype=torch.float16 if r16 else torch.float32)
        rt[..., :C_r] = r.permute(0, 2, 3, 4, 1).to(rt.dtype)
        res_t = rt.to(DEV)
        res_flags |= L.PTX_RES_F16 if r16 else 0
    raw_want = v
    ld_aff = _r4(Co) + 4
    sc = torch.rand(N, ld_aff, generator=torch.Generator().manual_seed(305)) + 0.5
    sh = rnd(N, ld_aff, seed=306, scale=0.3)
    if affine:
        v = v * sc[:, :Co, None, None, None] + sh[:, :Co, None, None, None]
    v = F.relu(v) if relu else v
    v = torch.tanh(v) if tanh else v
    # operands on the device
    ldh = (Ci + 7) // 8 * 8
    xh = torch.zeros(N, 1, H, W, ldh, dtype=torch.float16)
    xh[..., :Ci] = x.permute(0, 2, 3, 4, 1).half()
    xd = xh.to(DEV)
    pd = L.PackDesc(Co, Ci, 1, k, k, ldh, (Co + 127) // 128 * 128, 0, 0, 0, 0, 0, 0, 1)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV, dtype=torch.float16)
    bp = torch.empty(pd.Co_pad, device=DEV)
    wd, bd, scd, shd = w.to(DEV), bias.to(DEV), sc.to(DEV), sh.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bd), None, None, None, None, C.c_float(0),
                                     C.c_void_p(wp.data_ptr()), _p(bp), _st()), "pack f16")
    ldy = (Co + 7) // 8 * 8 if out16 else _r4(Co)
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, 1, Ho, Wo, Ci // 2, ldh // 2      # (Hi, Wi): the UPSAMPLED extents under up2
    d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, Ho, Wo, Co, ldy
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 1, k, k, 1, 1, 1, 0, pad, pad
    d.Kc, d.Co_pad, d.groups, d.ldr = ldh // 2, pd.Co_pad, 1, res_ld
    d.flags = (L.PTX_F16_OPERANDS | res_flags | (L.PTX_PRO_UP2 if up2 else 0) | (L.PTX_EPI_AFFINE if affine else 0) |
               (L.PTX_EPI_RELU if relu else 0) | (L.PTX_EPI_TANH if tanh else 0) | (L.PTX_EPI_OUT_F16 if out16 else 0) |
               (L.PTX_EPI_DUAL_RAW if dual else 0))
    ld_raw = (_r4(Co) + 7) // 8 * 8
    ext = L.ConvFusedExt()
    ext.scale, ext.shift, ext.ld_affine = scd.data_ptr(), shd.data_ptr(), ld_aff
    ws = torch.empty(4 * N * Ho * Wo * _r4(Co), device=DEV)
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    tol_out = 2e-3 if out16 else 2e-4                       # halfs out: 2^-11 relative rounding of the stored value
    for cfg, split in cfgs:
        yd = torch.full((N, 1, Ho, Wo, ldy), float("nan"), device=DEV, dtype=torch.float16 if out16 else torch.float32)
        rawd = torch.full((N, 1, Ho, Wo, ld_raw), float("nan"), device=DEV, dtype=torch.float16)
        ext.y_raw, ext.ld_raw = rawd.data_ptr(), ld_raw
        L.check(lib.ptx_conv3d_fused_fwd(C.byref(d), C.c_void_p(xd.data_ptr()), C.c_void_p(wp.data_ptr()), _p(bp),
                                         C.c_void_p(res_t.data_ptr()) if res_t is not None else None,
                                         C.c_void_p(yd.data_ptr()), C.byref(ext), _p(ws), ws.numel() * 4, cfg, split, _st()),
                "fused conv cfg %s" % (names[cfg] if cfg >= 0 else "auto"))
        torch.cuda.synchronize()
        got = yd.float().cpu()[..., :Co].permute(0, 4, 1, 2, 3)
        close(got, v, tol=tol_out)
        if dual:
            close(rawd.float().cpu()[..., :Co].permute(0, 4, 1, 2, 3), raw_want, tol=2e-3)


def test_conv_fused_generator_stage(ptx):
    """BASELINE.json config 5's "fused cBN + upsample + conv generator stage": every piece of
    ptx_conv3d_fused_fwd's epilogue / loader, each on several fp16 tiles (and split-K through the reduce kernel)."""
    lib = _lib(ptx)
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    f16 = [i for i, n in enumerate(names) if n.endswith("/f16") and "/kwr/" not in n]
    wide = [(c, 1) for c in f16 if int(names[c].split("x")[1]) >= 32]
    every = [(-1, 0)] + wide + [(wide[0][0], 3)]
    some = [(-1, 0), wide[1], wide[-1], (wide[2][0], 2)]
    # conv1 of a GBlock: 1x1, next cBN + ReLU, halfs out
    _fused_stage_case(ptx, 3, 6, 6, 64, 32, 1, False, True, True, False, True, False, None, every)
    # conv2 of an upsampling GBlock: the loader upsamples; several samples inside one 32-row tile (4x4 maps)
    _fused_stage_case(ptx, 2, 7, 5, 32, 32, 3, True, True, True, False, True, False, None, every)
    _fused_stage_case(ptx, 5, 2, 2, 48, 40, 3, True, True, True, False, True, False, None, some)
    # conv4: + upsampled, channel-truncated half skip, dual output (next block's input and skip)
    _fused_stage_case(ptx, 2, 8, 8, 32, 64, 1, False, True, True, False, True, True, ("up", True), every)
    _fused_stage_case(ptx, 2, 6, 6, 32, 64, 1, False, True, True, False, True, True, ("same", True), some)
    # first block after the linear layer / attention: fp32 skip operand; block before attention: raw fp32 out
    _fused_stage_case(ptx, 2, 8, 8, 32, 64, 1, False, True, True, False, True, True, ("up", False), some)
    _fused_stage_case(ptx, 2, 6, 6, 32, 64, 1, False, False, False, False, False, False, ("same", False), some)
    # image conv: 3 channels, tanh, fp32 out, 16-wide N tiles
    narrow = [(c, 1) for c in f16 if int(names[c].split("x")[1]) <= 32]
    _fused_stage_case(ptx, 2, 9, 9, 64, 3, 3, False, False, False, True, False, False, None, [(-1, 0)] + narrow)


def test_conv_x3_split_operands(ptx):
    """PTX_F16X3_OPERANDS: fp32 operands split into (hi, lo) halfs, three fp16 MFMAs, fp32 accumulate -- must agree
    with the fp32 reference to fp32-class accuracy (NOT fp16 accuracy): bound 2e-5 of the output scale, ten times
    tighter than the bar of the fp32-MFMA tiles and ~200x below what plain fp16 operands give (4e-3)."""
    lib = _lib(ptx)
    cfgs = x3_configs(lib)
    assert len(cfgs) >= 8
    worst = 0.0
    # every x3 tile x split-K on a 3x3x3 conv with residual + BN + ReLU
    N, T, H, W, Ci, Co = 2, 3, 9, 10, 64, 160
    x, w = rnd(N, Ci, T, H, W, seed=4), rnd(Co, Ci, 3, 3, 3, seed=5, scale=0.03)
    bn, res = make_bn(Co, 6), rnd(N, Co, T, H, W, seed=7)
    want = ref_conv(x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, res=res)
    for cfg in cfgs:
        for split in (1, 2, 5):
            got = hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, res=res, cfg=cfg, split=split, x3=True)
            err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
            worst = max(worst, err)
            assert err <= 2e-5, (lib.ptx_conv3d_config_name(cfg).decode(), split, err)
    # ragged channel counts (K tails inside an 8-channel block, N tails), strides, 1x1x1, temporal filters
    for g in GEOMS:
        name, N, T, H, W, Ci, Co, k, s_, p = g
        x, w = rnd(N, Ci, T, H, W, seed=1), rnd(Co, Ci, *k, seed=2, scale=(Ci * k[0] * k[1] * k[2]) ** -0.5)
        bn = make_bn(Co, 3)
        want = ref_conv(x, w, s_, p, bn=bn, relu=True)
        for cfg in [-1] + cfgs:
            got = hip_conv(ptx, x, w, s_, p, bn=bn, relu=True, cfg=cfg, split=1, x3=True)
            err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
            worst = max(worst, err)
            assert err <= 2e-5, (name, cfg, err)
    # wide dynamic range: large and tiny magnitudes in one reduction (the lo halfs of tiny values go subnormal:
    # absolute, not relative, accuracy is what the split guarantees)
    x = rnd(1, 64, 2, 8, 8, seed=30) * torch.logspace(-3, 2, 64).view(1, 64, 1, 1, 1)
    w = rnd(64, 64, 1, 3, 3, seed=31, scale=0.05)
    want = ref_conv(x, w, (1, 1, 1), (0, 1, 1))
    got = hip_conv(ptx, x, w, (1, 1, 1), (0, 1, 1), x3=True)
    assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    # an fp32 tile may not run a split problem, and vice versa
    L = ptx._lib
    with pytest.raises(L.PtxError):
        hip_conv(ptx, x, w, (1, 1, 1), (0, 1, 1), cfg=24, x3=True)
    with pytest.raises(L.PtxError):
        hip_conv(ptx, x, w, (1, 1, 1), (0, 1, 1), cfg=cfgs[0])
    print("x3 worst relative error %.2e" % worst)


def test_x3_dual_source_and_stem(ptx):
    """Split operands through the two other conv entry shapes of the ResNet3D plan: the K-concatenated
    conv3 + shortcut-B GEMM (ptx_conv3d_dual_fwd) and the kW-folded stem on 32-float rows."""
    L, lib = ptx._lib, _lib(ptx)
    null = C.c_void_p(0)
    r8 = lambda v: (v + 7) // 8 * 8      # noqa: E731
    for (N, T, H, W, C1, C2, Co, s_) in [(2, 4, 8, 8, 64, 64, 256, 1), (2, 2, 7, 7, 128, 256, 512, 2), (1, 1, 5, 6, 20, 36, 72, 2)]:
        T2, H2, W2 = (T - 1) * s_ + 1, (H - 1) * s_ + 1 + (s_ - 1), (W - 1) * s_ + 1
        o, x = rnd(N, C1, T, H, W, seed=60), rnd(N, C2, T2, H2, W2, seed=61)
        w3, wd = rnd(Co, C1, 1, 1, 1, seed=62, scale=C1 ** -0.5), rnd(Co, C2, 1, 1, 1, seed=63, scale=C2 ** -0.5)
        bn3, bnd = make_bn(Co, 64), make_bn(Co, 65)
        want = F.relu(ref_conv(o, w3, (1, 1, 1), (0, 0, 0), bn=bn3) + ref_conv(x, wd, (s_, s_, s_), (0, 0, 0), bn=bnd))
        Kc, Kc2, Co_pad = r8(C1), r8(C2), (Co + 127) // 128 * 128
        ld = Kc + Kc2
        wp = torch.zeros((Co_pad * ld,), device=DEV)
        bp = torch.full((Co_pad,), float("nan"), device=DEV)
        for (wt, bn, Ci_, kc, koff, acc) in ((w3, bn3, C1, Kc, 0, 0), (wd, bnd, C2, Kc2, Kc, 1)):
            pd = L.PackDesc(Co, Ci_, 1, 1, 1, kc, Co_pad, 0, ld, koff, acc)
            pd.f16 = 2
            ts = [t.to(DEV) for t in bn[:4]]
            wdv = wt.contiguous().to(DEV)
            L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wdv), null, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                             C.c_float(1e-5), _p(wp), _p(bp), _st()), "pack dual")
            torch.cuda.synchronize()
        od, xd = to_cl(o), to_cl(x)
        ldy = _r4(Co)
        for cfg, split in [(-1, 0)] + [(c, 1) for c in x3_configs(lib)[:6]] + [(x3_configs(lib)[3], 2)]:
            yd = torch.full((N, T, H, W, ldy), float("nan"), device=DEV)
            d = L.ConvDesc()
            d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, C1, od.shape[-1]
            d.To, d.Ho, d.Wo, d.Co, d.ldy = T, H, W, Co, ldy
            d.kT = d.kH = d.kW = d.sT = d.sH = d.sW = 1
            d.Kc, d.Co_pad, d.flags = Kc, Co_pad, L.PTX_EPI_RELU | L.PTX_F16X3_OPERANDS
            d.x2_C, d.x2_ld, d.x2_T, d.x2_H, d.x2_W = C2, xd.shape[-1], T2, H2, W2
            d.x2_sT = d.x2_sH = d.x2_sW = s_
            ws_bytes = lib.ptx_conv3d_workspace_bytes(C.byref(d), 8)
            ws = torch.empty(max(ws_bytes // 4, 4), device=DEV)
            L.check(lib.ptx_conv3d_dual_fwd(C.byref(d), _p(od), _p(xd), _p(wp), _p(bp), _p(yd), _p(ws), ws_bytes, cfg,
                                            split, _st()), "dual conv x3")
            torch.cuda.synchronize()
            close(from_cl(yd, Co), want, tol=2e-5)
    # stem: fold to 32-float rows, split filter, (7,7,1) conv
    N, T, H, W = 2, 5, 30, 26
    x, w = rnd(N, 3, T, H, W, seed=20), rnd(64, 3, 7, 7, 7, seed=21, scale=0.03)
    bn = make_bn(64, 22)
    want = ref_conv(x, w, (1, 2, 2), (3, 3, 3), bn=bn, relu=True)
    Wo, Ho = (W + 6 - 7) // 2 + 1, (H + 6 - 7) // 2 + 1
    xd = x.to(DEV)
    x2 = torch.full((N, T, H, Wo, 32), float("nan"), device=DEV)
    L.check(lib.ptx_fold_kw_ncdhw(_p(xd), _p(x2), N, 3, T, H, W, 7, 2, 3, Wo, 32, _st()), "fold")
    pd = L.PackDesc(64, 3, 7, 7, 7, 32, 128, 1)
    pd.f16 = 2
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(128, device=DEV)
    ts = [t.to(DEV) for t in bn[:4]]
    wd = w.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), None, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                     C.c_float(1e-5), _p(wp), _p(bp), _st()), "pack")
    for cfg in [-1] + x3_configs(lib):
        if cfg >= 0 and lib.ptx_conv3d_config_name(cfg).decode().split("/")[0].split("x")[1] not in ("64", "32"):
            continue
        yd = torch.full((N, T, Ho, Wo, 64), float("nan"), device=DEV)
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, Wo, 21, 32
        d.To, d.Ho, d.Wo, d.Co, d.ldy = T, Ho, Wo, 64, 64
        d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 7, 7, 1, 1, 2, 1, 3, 3, 0
        d.Kc, d.Co_pad, d.flags = 32, 128, L.PTX_EPI_RELU | L.PTX_F16X3_OPERANDS
        L.check(lib.ptx_conv3d_fwd(C.byref(d), _p(x2), _p(wp), _p(bp), None, _p(yd), None, 0, cfg, 1, _st()), "conv")
        torch.cuda.synchronize()
        close(from_cl(yd, 64), want, tol=2e-5)


def test_conv_kwr_tiles(ptx):
    """kw-reuse tiles: the A tile of a 3-wide stride-1 filter is staged once per (kt, kh, channel chunk) as the halo'd
    input run and serves the three kw taps.  Same results as every other tile of their operand flavour, on geometries
    whose M tile is a whole number of output rows; anything else is refused (PTX_ERR_UNSUPPORTED), never mis-computed."""
    L, lib = ptx._lib, _lib(ptx)
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    kx3, kf16 = kwr_configs(lib, "x3"), kwr_configs(lib, "f16")
    assert len(kx3) >= 4 and len(kf16) >= 4
    ran = 0
    # split operands: 3x3x3 and (1,3,3), W = 16 / 32 / 8, strides in T / H, ragged channels, M tails, residual
    for (N, T, H, W, Ci, Co, k, s_, p_) in [(2, 3, 6, 16, 64, 96, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
                                           (1, 4, 7, 32, 40, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                                           (3, 2, 9, 8, 64, 130, (3, 3, 3), (1, 2, 1), (1, 1, 1)),
                                           (2, 5, 5, 16, 128, 64, (3, 3, 3), (2, 1, 1), (1, 1, 1)),
                                           (1, 1, 7, 56, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1))]:
        x, w = rnd(N, Ci, T, H, W, seed=70), rnd(Co, Ci, *k, seed=71, scale=(Ci * k[0] * k[1] * k[2]) ** -0.5)
        bn = make_bn(Co, 72)
        To, Ho = (T + 2 * p_[0] - k[0]) // s_[0] + 1, (H + 2 * p_[1] - k[1]) // s_[1] + 1
        res = rnd(N, Co, To, Ho, W, seed=73)
        want = ref_conv(x, w, s_, p_, bn=bn, relu=True, res=res)
        for cfg in kx3:
            bm = int(names[cfg].split("x")[0])
            for split in (1, 2):
                if bm % W:
                    with pytest.raises(L.PtxError):
                        hip_conv(ptx, x, w, s_, p_, bn=bn, relu=True, res=res, cfg=cfg, split=1, x3=True)
                    break
                got = hip_conv(ptx, x, w, s_, p_, bn=bn, relu=True, res=res, cfg=cfg, split=split, x3=True)
                err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
                assert err <= 2e-5, (names[cfg], (N, T, H, W, Ci, Co, k, s_), split, err)
                ran += 1
    assert ran >= 20
    # a filter that is not 3 wide / not stride 1 in W is refused
    x, w = rnd(1, 64, 1, 8, 16, seed=74), rnd(64, 64, 1, 1, 1, seed=75, scale=0.1)
    with pytest.raises(L.PtxError):
        hip_conv(ptx, x, w, (1, 1, 1), (0, 0, 0), cfg=kx3[0], x3=True)
    # fp16 operands through the fused generator stage: plain 3x3, the upsampling loader, the 3-channel tanh image conv
    wide = [(c, 1) for c in kf16 if int(names[c].split("x")[1]) >= 32]
    narrow = [(c, 1) for c in kf16 if int(names[c].split("x")[1]) < 32]
    assert wide and narrow
    _fused_stage_case(ptx, 2, 8, 16, 64, 64, 3, False, True, True, False, True, False, None, wide + [(wide[0][0], 2)])
    _fused_stage_case(ptx, 2, 8, 8, 32, 32, 3, True, True, True, False, True, False, None, wide)          # up2: 16-wide output
    _fused_stage_case(ptx, 1, 16, 16, 48, 40, 3, True, True, True, False, True, False, None, wide)        # up2: 32-wide output
    _fused_stage_case(ptx, 2, 16, 32, 64, 3, 3, False, False, False, True, False, False, None, narrow + wide[:1])


def test_splitk_fused_reduction(ptx):
    """PTX_SPLITK_FUSED: the last split block of a tile reduces in-kernel.  Bit-identical to the two-launch path (partials
    are summed in split order either way), the counters return to zero (one workspace serves launch after launch, across
    tile shapes), and every epilogue (BN bias, residual, shortcut-A, ReLU) goes through it."""
    lib = _lib(ptx)
    names = [lib.ptx_conv3d_config_name(i).decode() for i in range(lib.ptx_conv3d_num_configs())]
    N, T, H, W, Ci, Co = 2, 3, 9, 10, 64, 160
    x, w = rnd(N, Ci, T, H, W, seed=4), rnd(Co, Ci, 3, 3, 3, seed=5, scale=0.03)
    bn, res = make_bn(Co, 6), rnd(N, Co, T, H, W, seed=7)
    ws = torch.zeros(8 * N * T * H * W * Co + 65536 // 4 + 64, device=DEV)
    fp32 = [c for c in fp32_configs(lib) if not names[c].endswith("/direct")]
    for cfg in fp32[::3] + x3_configs(lib)[::4]:
        is_x3 = names[cfg].endswith("/x3")
        for split in (2, 5, 8):
            base = hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, res=res, cfg=cfg, split=split, x3=is_x3)
            for rep_ in range(2):
                got = hip_conv(ptx, x, w, (1, 1, 1), (1, 1, 1), bn=bn, relu=True, res=res, cfg=cfg, split=split, x3=is_x3,
                               fused_split=ws)
                assert torch.equal(got, base), (names[cfg], split, rep_)
            assert int(ws[:16384].abs().sum().item()) == 0, "tile counters must return to zero"
    # shortcut-A residual through the fused tail, strided conv
    xs, w3 = rnd(2, 64, 4, 8, 8, seed=11), rnd(128, 64, 3, 3, 3, seed=16, scale=0.03)
    rp2, bn2 = rnd(2, 48, 4, 8, 8, seed=17), make_bn(128, 13)
    base = hip_conv(ptx, xs, w3, (2, 2, 2), (1, 1, 1), bn=bn2, relu=True, res_pad=rp2, res_stride=2, split=3)
    got = hip_conv(ptx, xs, w3, (2, 2, 2), (1, 1, 1), bn=bn2, relu=True, res_pad=rp2, res_stride=2, split=3, fused_split=ws)
    assert torch.equal(got, base)


def test_conv_stem_x3_direct(ptx):
    """ptx_conv_stem_x3_fwd: small-Cin stems with split operands, read from 4-channel (16-byte) positions -- against
    Conv3d / Conv2d + BN + ReLU in torch fp32.  ResNet3D stem (7^3, stride (1,2,2)), the 2-D ResNet stem, the (1,7,7)
    spatial stem of R2Plus1D with its 110 output channels (two N tiles, ragged), a strided-in-time I3D-like stem, and the
    MNIST net's single-channel 3x3."""
    L, lib = ptx._lib, _lib(ptx)
    cases = [  # N, Cin, T, H, W, Co, (kT,kH,kW), stride, pad
        (2, 3, 5, 30, 26, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),
        (3, 3, 1, 33, 40, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
        (2, 3, 4, 20, 18, 110, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
        (1, 3, 9, 18, 22, 64, (7, 7, 7), (2, 2, 2), (3, 3, 3)),
        (4, 1, 1, 28, 28, 32, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
        (1, 3, 3, 224, 224, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3)),          # full 112-wide rows: R = 4 rows per workgroup
        (2, 3, 6, 21, 24, 64, (7, 7, 7), (2, 2, 2), "same"),               # TF-SAME (I3D Unit3D): front pad 2, back pad 3 / 2
        (2, 3, 2, 40, 48, 110, (1, 7, 7), (1, 2, 2), (0, 3, 3)),           # planar: two channel tiles, one temporal tap
        (1, 2, 3, 19, 32, 64, (3, 6, 5), (1, 1, 2), (1, 2, 2)),            # planar: even kH, 5-wide filter with an even pad, 2 channels
    ]
    n_planar = 0
    for (N, Ci, T, H, W, Co, k, s_, p_) in cases:
        x = rnd(N, Ci, T, H, W, seed=80)
        w = rnd(Co, Ci, *k, seed=81, scale=(Ci * k[0] * k[1] * k[2]) ** -0.5)
        bn = make_bn(Co, 82)
        if p_ == "same":                 # F.pad(front = total // 2, back = rest) then an unpadded conv, as I3D ports do
            outs = [-(-i // st) for i, st in zip((T, H, W), s_)]
            tot = [max((o - 1) * st + kk - i, 0) for o, st, kk, i in zip(outs, s_, k, (T, H, W))]
            p_ = tuple(t // 2 for t in tot)
            xp = F.pad(x, (tot[2] // 2, tot[2] - tot[2] // 2, tot[1] // 2, tot[1] - tot[1] // 2, tot[0] // 2, tot[0] - tot[0] // 2))
            want = ref_conv(xp, w, s_, (0, 0, 0), bn=bn, relu=True)
        else:
            want = ref_conv(x, w, s_, p_, bn=bn, relu=True)
        To, Ho, Wo = want.shape[2:]
        xsrc = x.contiguous().to(DEV)
        xd = torch.full((N, T, H, W, 4), float("nan"), device=DEV)        # 16-byte positions: (hi4 | lo4) halfs
        L.check(lib.ptx_ncdhw_to_split4(_p(xsrc), _p(xd), N, Ci, T * H * W, _st()), "split4")
        torch.cuda.synchronize()
        halves = xd.view(torch.float16).view(N, T, H, W, 8).float().cpu()
        back = (halves[..., :4] + halves[..., 4:] / 4096.0)[..., :Ci].permute(0, 4, 1, 2, 3)        # lo is stored scaled by 2^12
        assert (back - x).abs().max().item() <= 2.0 ** -21 * max(1.0, x.abs().max().item())      # hi + lo == v to 22 bits
        assert bool((halves[..., Ci:4] == 0).all()) and bool((halves[..., 4 + Ci:] == 0).all())
        w4 = torch.zeros(Co, 4, *k)
        w4[:, :Ci] = w
        Co_pad = (Co + 127) // 128 * 128
        pd = L.PackDesc(Co, 4, k[0], k[1], k[2], 32, Co_pad, 1)
        pd.f16 = 2
        wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
        bp = torch.empty(Co_pad, device=DEV)
        ts = [t.to(DEV) for t in bn[:4]]
        wd = w4.contiguous().to(DEV)
        L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), None, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                         C.c_float(bn[4]), _p(wp), _p(bp), _st()), "pack stem")
        ldy = _r4(Co)
        yd = torch.full((N, To, Ho, Wo, ldy), float("nan"), device=DEV)
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, 4
        d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, Co, ldy
        d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = k[0], k[1], k[2], s_[0], s_[1], s_[2], p_[0], p_[1], p_[2]
        d.Kc, d.Co_pad, d.flags = 32, Co_pad, L.PTX_EPI_RELU | L.PTX_F16X3_OPERANDS
        assert lib.ptx_conv_stem_x3_supported(C.byref(d)), (N, Ci, T, H, W)
        L.check(lib.ptx_conv_stem_x3_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), _p(yd), _st()), "stem x3")
        torch.cuda.synchronize()
        got = from_cl(yd, Co)
        err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
        assert err <= 2e-5, ((N, Ci, T, H, W, Co, k, s_), err)
        pad = yd[..., Co:ldy]
        assert pad.numel() == 0 or bool((pad == 0).all())
        # the PLANAR kernel (six half planes per frame, 3 operands per filter row): same descriptor, same packed filter
        # (re-laid by ptx_pack_stem_x3p_weight), same products in a different summation order
        planar_ok = bool(lib.ptx_conv_stem_x3p_supported(C.byref(d)))
        assert planar_ok == (Ci <= 3 and s_[2] == 2 and W % 8 == 0 and k[2] <= 7), (N, Ci, T, H, W, k, s_)
        if planar_ok:
            n_planar += 1
            xpl = torch.full((N, T, 6, H, W), float("nan"), device=DEV, dtype=torch.float16)
            L.check(lib.ptx_ncdhw_to_split_planes(_p(xsrc), C.c_void_p(xpl.data_ptr()), N, Ci, T, H, W, _st()), "split planes")
            torch.cuda.synchronize()
            pl = xpl.float().cpu()
            back = (pl[:, :, :3] + pl[:, :, 3:] / 4096.0)[:, :, :Ci].permute(0, 2, 1, 3, 4)
            assert (back - x).abs().max().item() <= 2.0 ** -21 * max(1.0, x.abs().max().item())
            assert bool((pl[:, :, Ci:3] == 0).all()) and bool((pl[:, :, 3 + Ci:] == 0).all())
            wq = torch.full((lib.ptx_stem_x3p_weight_elems(C.byref(d)),), float("nan"), device=DEV)
            L.check(lib.ptx_pack_stem_x3p_weight(C.byref(d), _p(wp), _p(wq), _st()), "pack planar stem")
            yq = torch.full((N, To, Ho, Wo, ldy), float("nan"), device=DEV)
            L.check(lib.ptx_conv_stem_x3p_fwd(C.byref(d), C.c_void_p(xpl.data_ptr()), _p(wq), _p(bp), _p(yq), _st()), "stem x3 planar")
            torch.cuda.synchronize()
            assert not torch.isnan(wq).any()
            gq = from_cl(yq, Co)
            errq = (gq - want).abs().max().item() / max(1.0, want.abs().max().item())
            assert errq <= 2e-5, ("planar", (N, Ci, T, H, W, Co, k, s_), errq)
            padq = yq[..., Co:ldy]
            assert padq.numel() == 0 or bool((padq == 0).all())
    assert n_planar >= 3
    # refused, not mis-computed: fp32 operands, a 64-channel input, a 16-wide filter
    d.flags = L.PTX_EPI_RELU
    assert not lib.ptx_conv_stem_x3_supported(C.byref(d))
    assert lib.ptx_conv_stem_x3_fwd(C.byref(d), _p(xd), _p(wp), _p(bp), _p(yd), _st()) == 2


def test_conv_stem_f32_direct(ptx):
    """ptx_conv_stem_f32_fwd: the RGB stem on the fp32 matrix cores, read straight from the NCDHW tensor (no fold, no
    layout pass) -- against Conv3d / Conv2d + BN + ReLU in torch fp32.  ResNet3D stem (7^3, stride (1,2,2)), the 2-D ResNet
    stem, the (1,7,7) spatial stem of R2Plus1D with 110 output channels (two channel tiles, ragged), a strided-in-time stem,
    full 112-wide rows (a 512-output span touching 6 rows), TF-SAME padding (I3D), and a frame-strided view
    (`input[:, :, ::2]`) passed as strides."""
    L, lib = ptx._lib, _lib(ptx)
    cases = [  # N, T, H, W, Co, (kT,kH,kW), stride, pad, frame step
        (2, 5, 30, 28, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3), 1),
        (3, 1, 33, 40, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3), 1),
        (2, 4, 20, 20, 110, (1, 7, 7), (1, 2, 2), (0, 3, 3), 1),
        (1, 9, 18, 24, 64, (7, 7, 7), (2, 2, 2), (3, 3, 3), 1),
        (1, 3, 224, 224, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), 1),
        (2, 6, 21, 24, 64, (7, 7, 7), (2, 2, 2), "same", 1),
        (2, 4, 26, 36, 64, (3, 7, 7), (1, 1, 1), (1, 3, 3), 2),             # stride-1 windows, every other frame of 8
        (8, 5, 30, 28, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3), 1),             # 8 (n, band) groups: longest-first tile order per XCD
        (4, 6, 40, 40, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3), 1),             # two 256-output bands per frame, 3 tap classes
    ]
    for (N, T, H, W, Co, k, s_, p_, step) in cases:
        full = rnd(N, 3, T * step, H, W, seed=180)
        x = full[:, :, ::step]
        w = rnd(Co, 3, *k, seed=181, scale=(3 * k[0] * k[1] * k[2]) ** -0.5)
        bn = make_bn(Co, 182)
        if p_ == "same":
            outs = [-(-i // st) for i, st in zip((T, H, W), s_)]
            tot = [max((o - 1) * st + kk - i, 0) for o, st, kk, i in zip(outs, s_, k, (T, H, W))]
            p_ = tuple(t // 2 for t in tot)
            xp = F.pad(x, (tot[2] // 2, tot[2] - tot[2] // 2, tot[1] // 2, tot[1] - tot[1] // 2, tot[0] // 2, tot[0] - tot[0] // 2))
            want = ref_conv(xp, w, s_, (0, 0, 0), bn=bn, relu=True)
        else:
            want = ref_conv(x, w, s_, p_, bn=bn, relu=True)
        To, Ho, Wo = want.shape[2:]
        xd = full.contiguous().to(DEV)
        Co_pad = (Co + 127) // 128 * 128
        Kc = 24
        pd = L.PackDesc(Co, 3, k[0], k[1], k[2], Kc, Co_pad, 1)
        wf = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
        bp = torch.empty(Co_pad, device=DEV)
        ts = [t.to(DEV) for t in bn[:4]]
        wd = w.contiguous().to(DEV)
        L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), None, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                         C.c_float(bn[4]), _p(wf), _p(bp), _st()), "pack folded")
        ldy = _r4(Co)
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, 3, 0
        d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, Co, ldy
        d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = k[0], k[1], k[2], s_[0], s_[1], s_[2], p_[0], p_[1], p_[2]
        d.Kc, d.Co_pad, d.flags = Kc, Co_pad, L.PTX_EPI_RELU
        plane = H * W
        sn, sc, st = 3 * T * step * plane, T * step * plane, step * plane
        assert lib.ptx_conv_stem_f32_supported(C.byref(d), sn, sc, st), (N, T, H, W)
        ws = torch.full((lib.ptx_stem_f32_weight_elems(C.byref(d)),), float("nan"), device=DEV)
        L.check(lib.ptx_pack_stem_f32_weight(C.byref(d), _p(wf), Kc, _p(ws), _st()), "pack stem f32")
        yd = torch.full((N, To, Ho, Wo, ldy), float("nan"), device=DEV)
        L.check(lib.ptx_conv_stem_f32_fwd(C.byref(d), _p(xd), sn, sc, st, _p(ws), _p(bp), _p(yd), _st()), "stem f32")
        torch.cuda.synchronize()
        got = from_cl(yd, Co)
        err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
        assert err <= 2e-5, ((N, T, H, W, Co, k, s_, step), err)
        pad = yd[..., Co:ldy]
        assert pad.numel() == 0 or bool((pad == 0).all())
    # refused, not mis-computed: a residual epilogue, a width that is not a multiple of 4, a 5-wide window
    d.flags = L.PTX_EPI_RELU | L.PTX_EPI_RES_ADD
    assert not lib.ptx_conv_stem_f32_supported(C.byref(d), sn, sc, st)
    assert lib.ptx_conv_stem_f32_fwd(C.byref(d), _p(xd), sn, sc, st, _p(ws), _p(bp), _p(yd), _st()) == 2
    d.flags, d.kW = L.PTX_EPI_RELU, 5
    assert not lib.ptx_conv_stem_f32_supported(C.byref(d), sn, sc, st)
    d.kW, d.Wi = 7, 38
    assert not lib.ptx_conv_stem_f32_supported(C.byref(d), sn, sc, st)


def test_conv_stem_f32_padded_pitch(ptx):
    """Widths that are not multiples of 4 floats: ptx_pad_rows copies the NCDHW rows to a zero-padded 16-byte pitch and
    ptx_conv_stem_f32_fwd reads them with desc.ldx = pitch (the reference takes any T/H/W, torchvision_models.py:448)."""
    L, lib = ptx._lib, _lib(ptx)
    for (N, T, H, W, Co, k, s_, p_) in [(2, 4, 18, 30, 64, (3, 7, 7), (1, 2, 2), (1, 3, 3)),
                                         (1, 1, 33, 45, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),
                                         (2, 5, 20, 22, 110, (1, 7, 7), (1, 2, 2), (0, 3, 3))]:
        x = rnd(N, 3, T, H, W, seed=190)
        w = rnd(Co, 3, *k, seed=191, scale=(3 * k[0] * k[1] * k[2]) ** -0.5)
        bn = make_bn(Co, 192)
        want = ref_conv(x, w, s_, p_, bn=bn, relu=True)
        To, Ho, Wo = want.shape[2:]
        pitch = _r4(W)
        xd = x.contiguous().to(DEV)
        xp = torch.full((N * 3 * T * H, pitch), float("nan"), device=DEV)
        L.check(lib.ptx_pad_rows(_p(xd), _p(xp), N * 3 * T * H, W, pitch, _st()), "pad_rows")
        torch.cuda.synchronize()
        xpc = xp.cpu().reshape(N, 3, T, H, pitch)
        assert torch.equal(xpc[..., :W], x) and bool((xpc[..., W:] == 0).all())
        Co_pad = (Co + 127) // 128 * 128
        Kc = 24
        pd = L.PackDesc(Co, 3, k[0], k[1], k[2], Kc, Co_pad, 1)
        wf = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
        bp = torch.empty(Co_pad, device=DEV)
        ts = [t.to(DEV) for t in bn[:4]]
        wd = w.contiguous().to(DEV)
        L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), None, _p(ts[0]), _p(ts[1]), _p(ts[2]), _p(ts[3]),
                                         C.c_float(bn[4]), _p(wf), _p(bp), _st()), "pack folded")
        ldy = _r4(Co)
        d = L.ConvDesc()
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, 3, pitch
        d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, Co, ldy
        d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = k[0], k[1], k[2], s_[0], s_[1], s_[2], p_[0], p_[1], p_[2]
        d.Kc, d.Co_pad, d.flags = Kc, Co_pad, L.PTX_EPI_RELU
        plane = H * pitch
        sn, sc, st = 3 * T * plane, T * plane, plane
        assert lib.ptx_conv_stem_f32_supported(C.byref(d), sn, sc, st), (N, T, H, W)
        ws = torch.full((lib.ptx_stem_f32_weight_elems(C.byref(d)),), float("nan"), device=DEV)
        L.check(lib.ptx_pack_stem_f32_weight(C.byref(d), _p(wf), Kc, _p(ws), _st()), "pack stem f32")
        yd = torch.full((N, To, Ho, Wo, ldy), float("nan"), device=DEV)
        L.check(lib.ptx_conv_stem_f32_fwd(C.byref(d), _p(xp), sn, sc, st, _p(ws), _p(bp), _p(yd), _st()), "stem f32 (pitch)")
        torch.cuda.synchronize()
        got = from_cl(yd, Co)
        err = (got - want).abs().max().item() / max(1.0, want.abs().max().item())
        assert err <= 2e-5, ((N, T, H, W, Co), err)


def _pack_plain(ptx, w, bn, bias=None, x3=False):
    """BN-folded K-major packed filter + bias on the device (unfolded; fp32, or (hi8 | lo8) half blocks for x3)."""
    L, lib = ptx._lib, _lib(ptx)
    Co, Ci, kT, kH, kW = w.shape
    pd = L.PackDesc(Co, Ci, kT, kH, kW, (Ci + 7) // 8 * 8 if x3 else _r4(Ci), (Co + 127) // 128 * 128, 0)
    pd.f16 = 2 if x3 else 0
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(pd.Co_pad, device=DEV)
    wd = w.contiguous().to(DEV)
    null = C.c_void_p(0)
    bnargs, eps, keep = [null] * 4, 0.0, [wd]
    if bn is not None:
        ts = [t.contiguous().to(DEV) for t in bn[:4]]
        keep += ts
        bnargs, eps = [_p(t) for t in ts], bn[4]
    bd = bias.to(DEV) if bias is not None else None
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bd) if bd is not None else null, *bnargs, C.c_float(eps),
                                     _p(wp), _p(bp), _st()), "pack")
    torch.cuda.synchronize()
    return pd, wp, bp


CHAIN_CASES = [
    # name, N,T,H,W, Ci, N1, Co2, k, stride, pad, relu1, relu2, residual
    ("bottleneck tail 3x3x3 -> 1x1x1 + residual", 2, 4, 14, 15, 64, 64, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, True, True),
    ("(2+1)D pointwise pair 64 -> 51 -> 256 + residual", 2, 3, 13, 14, 64, 51, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, True, True),
    ("pointwise pair 256 -> 51 -> 64, ReLU out, no residual", 1, 4, 12, 12, 256, 51, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, True, False),
    ("strided pair 256 -> 85 -> 128 (composed stride 2,2,2), linear output", 2, 4, 12, 14, 256, 85, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0), True, False, False),
    ("narrow mid 64 -> 32 -> 64", 2, 2, 9, 11, 64, 32, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, True, False),
    ("mid 102 (128-wide N tile): 512 -> 102 -> 512 + residual", 1, 2, 10, 9, 512, 102, 512, (1, 1, 1), (1, 1, 1), (0, 0, 0), True, True, True),
    ("layer2 tail 3x3x3 stride 2 -> 1x1x1, 128 planes", 1, 4, 12, 12, 128, 128, 512, (3, 3, 3), (2, 2, 2), (1, 1, 1), True, True, False),
    ("(1,3,3) conv -> ragged tail 144 -> 100 -> 36, no inner ReLU", 1, 3, 10, 10, 144, 100, 36, (1, 3, 3), (1, 1, 1), (0, 1, 1), False, True, False),
    # 28-wide rows: the kw-reuse chained tiles (224 rows = 8 whole output rows; 9 rows per frame: tiles straddle frames)
    ("kw-reuse tail 3x3x3 -> 1x1x1 + residual, Wo 28", 1, 3, 9, 28, 64, 64, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1), True, True, True),
]


@pytest.mark.parametrize("kind", ["fp32", "x3"])
@pytest.mark.parametrize("case", CHAIN_CASES, ids=[c[0] for c in CHAIN_CASES])
def test_conv_chain(ptx, case, kind):
    """ptx_conv3d_chain_fwd (conv -> BN -> ReLU -> 1x1x1 conv -> BN -> + residual -> ReLU in one launch, the intermediate
    tile in LDS) on EVERY chained tile that holds the intermediate row, against (a) the op sequence in torch fp32 on the
    CPU and (b) the two launches it replaces on the plain tiles of the same MFMA shape -- bit for bit (same k order:
    VERDICT r2 #3 'bit-exactness test vs the unfused pair').  kind "x3": both GEMMs on split operands."""
    x3 = kind == "x3"
    L, lib = ptx._lib, _lib(ptx)
    _, N, T, H, W, Ci, N1, Co2, k, s_, p_, relu1, relu2, with_res = case
    x = rnd(N, Ci, T, H, W, seed=400 + Ci)
    w1 = rnd(N1, Ci, *k, seed=401, scale=(Ci * k[0] * k[1] * k[2]) ** -0.5)
    w2 = rnd(Co2, N1, 1, 1, 1, seed=402, scale=N1 ** -0.5)
    bn1, bn2 = make_bn(N1, 403), make_bn(Co2, 404)
    mid = ref_conv(x, w1, s_, p_, bn=bn1, relu=relu1)
    To, Ho, Wo = mid.shape[2:]
    res = rnd(N, Co2, To, Ho, Wo, seed=405) if with_res else None
    want = ref_conv(mid, w2, (1, 1, 1), (0, 0, 0), bn=bn2, relu=relu2, res=res)
    pd1, wp1, bp1 = _pack_plain(ptx, w1, bn1, x3=x3)
    pd2, wp2, bp2 = _pack_plain(ptx, w2, bn2, x3=x3)
    fx3 = L.PTX_F16X3_OPERANDS if x3 else 0
    xd = to_cl(x)
    rd = to_cl(res) if with_res else None
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, xd.shape[-1]
    d.To, d.Ho, d.Wo, d.Co, d.ldy = To, Ho, Wo, N1, _r4(N1)
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = k[0], k[1], k[2], s_[0], s_[1], s_[2], p_[0], p_[1], p_[2]
    d.Kc, d.Co_pad, d.flags = pd1.Kc, pd1.Co_pad, (L.PTX_EPI_RELU if relu1 else 0) | fx3
    ldy = _r4(Co2) + 8                                   # a row stride wider than the written columns
    d2 = L.ConvDesc()
    d2.N, d2.Ti, d2.Hi, d2.Wi, d2.Ci, d2.ldx = N, To, Ho, Wo, N1, _r4(N1)
    d2.To, d2.Ho, d2.Wo, d2.Co, d2.ldy = To, Ho, Wo, Co2, ldy
    d2.kT = d2.kH = d2.kW = d2.sT = d2.sH = d2.sW = 1
    d2.Kc, d2.Co_pad = pd2.Kc, pd2.Co_pad
    d2.flags = (L.PTX_EPI_RELU if relu2 else 0) | (L.PTX_EPI_RES_ADD if with_res else 0) | fx3
    d2.ldr = rd.shape[-1] if with_res else 0
    plain = {lib.ptx_conv3d_config_name(i).decode(): i for i in range(lib.ptx_conv3d_num_configs())}
    null = C.c_void_p(0)
    ran = pairs = 0
    for cfg in range(lib.ptx_conv3d_chain_num_configs()):
        name = lib.ptx_conv3d_chain_config_name(cfg).decode()
        tile_x3 = name.endswith("/x3")                               # ".../chain[/re]/x3": split-operand tiles
        core = name[:-3] if tile_x3 else name
        core = core[:-3] if core.endswith("/re") else core           # ".../chain/re": the tail's epilogue row-major through LDS
        assert core.endswith(("/dma/chain", "/dma/kwr/chain"))
        bm_tile, bn_tile = int(name.split("x")[0]), int(name.split("x")[1])
        # ".../kwr/chain": the first conv through the kw-reuse loader -- 3-wide stride-1 filters, tiles of whole output rows
        kwr_ok = "/kwr/" not in name or (k[2] == 3 and s_[2] == 1 and Wo >= 8 and bm_tile % Wo == 0 and Wo == W + 2 * p_[2] - 2)
        ok = lib.ptx_conv3d_chain_supported(C.byref(d), C.byref(d2), cfg)
        assert bool(ok) == (_r4(N1) <= bn_tile and tile_x3 == x3 and kwr_ok), (name, N1)
        if not ok:
            yd = torch.zeros((N, To, Ho, Wo, ldy), device=DEV)
            assert lib.ptx_conv3d_chain_fwd(C.byref(d), C.byref(d2), _p(xd), _p(wp1), _p(bp1), _p(wp2), _p(bp2),
                                            _p(rd) if with_res else null, _p(yd), cfg, _st()) == 2      # refused, not mis-computed
            continue
        yd = torch.full((N, To, Ho, Wo, ldy), float("nan"), device=DEV)
        L.check(lib.ptx_conv3d_chain_fwd(C.byref(d), C.byref(d2), _p(xd), _p(wp1), _p(bp1), _p(wp2), _p(bp2),
                                         _p(rd) if with_res else null, _p(yd), cfg, _st()), "chain " + name)
        torch.cuda.synchronize()
        got = from_cl(yd, Co2)
        close(got, want, 2e-5)
        assert bool((yd[..., Co2:_r4(Co2)] == 0).all()) and bool(torch.isnan(yd[..., _r4(Co2):]).all()), name   # pad cols zero, beyond untouched
        ran += 1
        base = core[:-len("/chain")] + ("/x3" if x3 else "")
        if base in plain:                                # the two launches it replaces, same tile / MFMA shape: bit-identical
            mid_g = hip_conv(ptx, x, w1, s_, p_, bn=bn1, relu=relu1, cfg=plain[base], split=1, x3=x3)
            # (a kw-reuse tile does not take the pointwise tail: any 32x32x16 split-operand tile sums k in the same order)
            tail_cfg = plain["64x64x32/2x2/m32/dma/x3"] if "/kwr/" in name else plain[base]
            two = hip_conv(ptx, mid_g, w2, (1, 1, 1), (0, 0, 0), bn=bn2, relu=relu2, res=res, cfg=tail_cfg, split=1, x3=x3)
            pairs += 1
            assert torch.equal(got, two), (name, (got - two).abs().max().item())
    assert ran >= 1 and pairs >= 1
    # config < 0: the library's own pick
    yd = torch.full((N, To, Ho, Wo, ldy), float("nan"), device=DEV)
    L.check(lib.ptx_conv3d_chain_fwd(C.byref(d), C.byref(d2), _p(xd), _p(wp1), _p(bp1), _p(wp2), _p(bp2),
                                     _p(rd) if with_res else null, _p(yd), -1, _st()), "chain auto")
    torch.cuda.synchronize()
    close(from_cl(yd, Co2), want, 2e-5)
    # refused: a residual on the FIRST conv, a non-pointwise tail, mismatched positions
    d.flags |= L.PTX_EPI_RES_ADD
    assert lib.ptx_conv3d_chain_fwd(C.byref(d), C.byref(d2), _p(xd), _p(wp1), _p(bp1), _p(wp2), _p(bp2), null, _p(yd), -1, _st()) != 0
    d.flags &= ~L.PTX_EPI_RES_ADD
    d2.Wi += 1
    assert lib.ptx_conv3d_chain_fwd(C.byref(d), C.byref(d2), _p(xd), _p(wp1), _p(bp1), _p(wp2), _p(bp2), null, _p(yd), -1, _st()) != 0
    d2.Wi -= 1
    d2.flags ^= L.PTX_F16X3_OPERANDS                     # split operands on one conv only
    assert lib.ptx_conv3d_chain_fwd(C.byref(d), C.byref(d2), _p(xd), _p(wp1), _p(bp1), _p(wp2), _p(bp2), null, _p(yd), -1, _st()) != 0


@pytest.mark.parametrize("N,H,W,Cc", [(2, 64, 64, 128), (3, 37, 70, 64), (1, 8, 32, 32), (2, 256, 256, 128)])
def test_rgb_conv3x3_f16(ptx, N, H, W, Cc):
    """ptx_rgb_conv3x3_f16_fwd (the generator's output layer, BN -> ReLU -> conv3x3(C -> 3) -> tanh in one launch: taps in
    the N axis of one GEMM over the input positions, shifted partial sums combined through LDS) against the op sequence in
    torch fp32 on the same half-rounded operands: per-sample affine + ReLU (rounded to half, as the fragment math does),
    zero padding of the ACTIVATED map, conv2d, tanh.  Ragged tiles, every compiled channel count, config 5's own extent."""
    L, lib = ptx._lib, _lib(ptx)
    g_ = torch.Generator().manual_seed(7 + H + Cc)
    ldx, lda = Cc + 8, Cc + 4
    x = (torch.randn(N, H, W, ldx, generator=g_) * 1.5).half()
    scale = torch.rand(N, lda, generator=g_) + 0.5
    shift = torch.randn(N, lda, generator=g_) * 0.3
    w = torch.randn(3, Cc, 3, 3, generator=g_) * (1.0 / (9 * Cc)) ** 0.5
    b = torch.randn(3, generator=g_) * 0.1
    act = F.relu((x[..., :Cc].float() * scale[:, None, None, :Cc].half().float() + shift[:, None, None, :Cc].half().float()).half().float())
    want = torch.tanh(F.conv2d(act.permute(0, 3, 1, 2), w.half().float(), b, padding=1)).permute(0, 2, 3, 1)
    xd, sd, hd, wd = x.to(DEV), scale.to(DEV), shift.to(DEV), w.to(DEV)
    bd = torch.zeros(4, device=DEV)
    bd[:3] = b.to(DEV)
    wp = torch.empty(int(lib.ptx_rgb_conv_weight_elems(Cc)), device=DEV, dtype=torch.float16)
    L.check(lib.ptx_pack_rgb_conv_weight(_p(wd), Cc, C.c_void_p(wp.data_ptr()), _st()), "pack")
    y = torch.full((N, H, W, 4), float("nan"), device=DEV)
    d = L.RgbConvDesc(N, H, W, Cc, ldx, 4, lda, L.PTX_EPI_TANH)
    assert lib.ptx_rgb_conv3x3_f16_supported(C.byref(d))
    L.check(lib.ptx_rgb_conv3x3_f16_fwd(C.byref(d), C.c_void_p(xd.data_ptr()), _p(sd), _p(hd), C.c_void_p(wp.data_ptr()), _p(bd), _p(y),
                                        _st()), "rgb conv")
    torch.cuda.synchronize()
    got = y.cpu()
    assert (got[..., 3] == 0).all()
    err = (got[..., :3] - want).abs().max().item()
    assert err <= 4e-3, (N, H, W, Cc, err)              # half-rounded affine tables + one fp16 rounding per activation
    d.C = 48
    assert not lib.ptx_rgb_conv3x3_f16_supported(C.byref(d))


@pytest.mark.parametrize("N,H,W,Cc,up2,affine,relu", [
    (2, 32, 64, 64, False, True, True),      # whole tiles
    (2, 16, 32, 64, True, True, True),       # upsampling loader: the filter slides over 32 x 64
    (1, 19, 45, 64, False, True, False),     # ragged tiles, no ReLU
    (2, 32, 32, 128, False, True, True),     # two 64-channel input chunks
    (1, 12, 20, 128, True, False, True),     # C = 128 + upsampling, no affine (bias only), ragged
    (3, 8, 32, 64, False, False, False),     # one tile per image, plain conv + bias
    (2, 32, 32, 256, False, True, True),     # 256 channels: four input chunks, output channels cut over blockIdx.y
    (1, 8, 16, 256, True, True, True),       # ... with the upsampling loader (16 x 32 output)
    (1, 19, 45, 64, True, True, True),       # ragged + upsampled
])
def test_conv3x3_f16_patch_kernel(ptx, N, H, W, Cc, up2, affine, relu):
    """ptx_conv3x3_f16_fwd (a GBlock's 3x3 conv from one staged input patch per 8 x 32 tile, gen_stage_f16.hip) against the op
    sequence in torch fp32 on the same half-rounded operands: [nearest x2] -> conv3x3 + bias -> per-sample affine -> ReLU ->
    halfs.  Also bit-compared in spirit with the implicit-GEMM fused stage it replaces (same operands, same packed filter):
    both must sit within the half-output rounding of the torch result."""
    L, lib = ptx._lib, _lib(ptx)
    x = rnd(N, Cc, 1, H, W, seed=400 + Cc + H).half().float()
    w = rnd(Cc, Cc, 1, 3, 3, seed=401, scale=(Cc * 9) ** -0.5).half().float()
    bias = rnd(Cc, seed=402)
    Ho, Wo = (2 * H, 2 * W) if up2 else (H, W)
    xin = F.interpolate(x[:, :, 0], scale_factor=2, mode="nearest").unsqueeze(2) if up2 else x
    v = F.conv3d(xin, w, bias, 1, (0, 1, 1))
    ld_aff = Cc + 4
    sc = torch.rand(N, ld_aff, generator=torch.Generator().manual_seed(405)) + 0.5
    sh = rnd(N, ld_aff, seed=406, scale=0.3)
    if affine:
        v = v * sc[:, :Cc, None, None, None] + sh[:, :Cc, None, None, None]
    v = F.relu(v) if relu else v
    want = v[:, :, 0].permute(0, 2, 3, 1)
    ldh = Cc + 8
    xh = torch.zeros(N, 1, H, W, ldh, dtype=torch.float16)
    xh[..., :Cc] = x.permute(0, 2, 3, 4, 1).half()
    xd = xh.to(DEV)
    pd = L.PackDesc(Cc, Cc, 1, 3, 3, ldh, (Cc + 127) // 128 * 128, 0, 0, 0, 0, 0, 0, 1)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV, dtype=torch.float16)
    bp = torch.empty(pd.Co_pad, device=DEV)
    wd, bd, scd, shd = w.to(DEV), bias.to(DEV), sc.to(DEV), sh.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bd), None, None, None, None, C.c_float(0),
                                     C.c_void_p(wp.data_ptr()), _p(bp), _st()), "pack f16")
    ldy = Cc + 16
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, 1, Ho, Wo, Cc // 2, ldh // 2
    d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, Ho, Wo, Cc, ldy
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 1, 3, 3, 1, 1, 1, 0, 1, 1
    d.Kc, d.Co_pad, d.groups = ldh // 2, pd.Co_pad, 1
    d.flags = (L.PTX_F16_OPERANDS | L.PTX_EPI_OUT_F16 | (L.PTX_PRO_UP2 if up2 else 0) | (L.PTX_EPI_AFFINE if affine else 0) |
               (L.PTX_EPI_RELU if relu else 0))
    assert lib.ptx_conv3x3_f16_supported(C.byref(d))
    ext = L.ConvFusedExt()
    ext.scale, ext.shift, ext.ld_affine = scd.data_ptr(), shd.data_ptr(), ld_aff
    yd = torch.full((N, 1, Ho, Wo, ldy), float("nan"), device=DEV, dtype=torch.float16)
    L.check(lib.ptx_conv3x3_f16_fwd(C.byref(d), C.c_void_p(xd.data_ptr()), C.c_void_p(wp.data_ptr()), _p(bp),
                                    C.c_void_p(yd.data_ptr()), C.byref(ext) if affine else None, _st()), "conv3x3_f16")
    torch.cuda.synchronize()
    got = yd.cpu().float()[:, 0]
    assert torch.isnan(got[..., Cc:]).all()                 # columns beyond Co are left untouched
    err = (got[..., :Cc] - want).abs().max().item()
    assert err <= 2e-3 * max(1.0, want.abs().max().item()), (N, H, W, Cc, up2, err)
    # what it refuses: strides, other widths, residuals
    d.sW = 2
    assert not lib.ptx_conv3x3_f16_supported(C.byref(d))
    d.sW, d.Ci = 1, 48
    assert not lib.ptx_conv3x3_f16_supported(C.byref(d))
    d.Ci, d.flags = Cc // 2, d.flags | L.PTX_EPI_RES_ADD
    assert not lib.ptx_conv3x3_f16_supported(C.byref(d))


@pytest.mark.parametrize("N,H,W,K,Co,skip,affine,dual,relu", [
    (2, 16, 32, 64, 128, "up", True, True, True),        # blocks.5.1-like: upsampled, channel-truncated skip, both outputs
    (2, 16, 64, 64, 256, "same", True, True, True),      # two output-channel parts over blockIdx.y, same-shape skip
    (1, 19, 45, 128, 128, "up", True, True, True),       # two input chunks, ragged tiles
    (1, 8, 32, 256, 256, "chan", True, True, False),     # four input chunks; skip = channel truncation only (log2 factor 0)
    (2, 8, 32, 64, 128, "up", False, False, False),      # the last block: raw output only
    (1, 12, 40, 128, 128, None, True, False, True),      # no skip, activated output only
])
def test_conv1x1_skip_f16_kernel(ptx, N, H, W, K, Co, skip, affine, dual, relu):
    """ptx_conv1x1_skip_f16_fwd (a GBlock's closing 1x1 conv: + skip, raw AND activated output) against torch fp32 on the same
    half-rounded operands; the raw output must be the single rounding of the fp32 sum."""
    L, lib = ptx._lib, _lib(ptx)
    x = rnd(N, K, 1, H, W, seed=500 + K + H).half().float()
    w = rnd(Co, K, 1, 1, 1, seed=501, scale=K ** -0.5).half().float()
    bias = rnd(Co, seed=502)
    v = F.conv3d(x, w, bias)
    d = L.ConvDesc()
    res_t = None
    if skip is not None:
        up = skip == "up"
        Cr = Co if skip == "same" else Co + 64
        rh, rw = (-(-H // 2), -(-W // 2)) if up else (H, W)
        r = rnd(N, Cr, 1, rh, rw, seed=503).half().float()
        rr = F.interpolate(r[:, :Co, 0], scale_factor=2, mode="nearest")[:, :, :H, :W].unsqueeze(2) if up else r[:, :Co]
        v = v + rr
        ldr = Cr + 8
        rt = torch.zeros(N, 1, rh, rw, ldr, dtype=torch.float16)
        rt[..., :Cr] = r.permute(0, 2, 3, 4, 1).half()
        res_t = rt.to(DEV)
        d.ldr = ldr
        if skip == "same":
            d.flags |= L.PTX_EPI_RES_ADD
        else:
            d.flags |= L.PTX_EPI_RES_PADA | L.PTX_EPI_RES_UP
            d.res_C, d.res_T, d.res_H, d.res_W = Cr, 1, rh, rw
            d.res_sT, d.res_sH, d.res_sW = 0, int(up), int(up)
        d.flags |= L.PTX_RES_F16
    raw_want = v[:, :, 0].permute(0, 2, 3, 1)
    ld_aff = Co + 4
    sc = torch.rand(N, ld_aff, generator=torch.Generator().manual_seed(505)) + 0.5
    sh = rnd(N, ld_aff, seed=506, scale=0.3)
    a = v * sc[:, :Co, None, None, None] + sh[:, :Co, None, None, None] if affine else v
    a = F.relu(a) if relu else a
    act_want = a[:, :, 0].permute(0, 2, 3, 1)
    ldh = K + 8
    xh = torch.zeros(N, 1, H, W, ldh, dtype=torch.float16)
    xh[..., :K] = x.permute(0, 2, 3, 4, 1).half()
    xd = xh.to(DEV)
    pd = L.PackDesc(Co, K, 1, 1, 1, ldh, (Co + 127) // 128 * 128, 0, 0, 0, 0, 0, 0, 1)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV, dtype=torch.float16)
    bp = torch.empty(pd.Co_pad, device=DEV)
    wd, bd, scd, shd = w.to(DEV), bias.to(DEV), sc.to(DEV), sh.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bd), None, None, None, None, C.c_float(0),
                                     C.c_void_p(wp.data_ptr()), _p(bp), _st()), "pack f16")
    ldy, ld_raw = Co + 8, Co + 16
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, 1, H, W, K // 2, ldh // 2
    d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, H, W, Co, ldy
    d.kT = d.kH = d.kW = d.sT = d.sH = d.sW = 1
    d.Kc, d.Co_pad, d.groups = ldh // 2, pd.Co_pad, 1
    d.flags |= (L.PTX_F16_OPERANDS | L.PTX_EPI_OUT_F16 | (L.PTX_EPI_AFFINE if affine else 0) | (L.PTX_EPI_RELU if relu else 0) |
                (L.PTX_EPI_DUAL_RAW if dual else 0))
    assert lib.ptx_conv1x1_skip_f16_supported(C.byref(d))
    yd = torch.full((N, 1, H, W, ldy), float("nan"), device=DEV, dtype=torch.float16)
    rawd = torch.full((N, 1, H, W, ld_raw), float("nan"), device=DEV, dtype=torch.float16)
    ext = L.ConvFusedExt()
    ext.scale, ext.shift, ext.ld_affine, ext.y_raw, ext.ld_raw = scd.data_ptr(), shd.data_ptr(), ld_aff, rawd.data_ptr(), ld_raw
    L.check(lib.ptx_conv1x1_skip_f16_fwd(C.byref(d), C.c_void_p(xd.data_ptr()), C.c_void_p(wp.data_ptr()), _p(bp),
                                         C.c_void_p(res_t.data_ptr()) if res_t is not None else None, C.c_void_p(yd.data_ptr()),
                                         C.byref(ext), _st()), "conv1x1_skip_f16")
    torch.cuda.synchronize()
    got = yd.cpu().float()[:, 0]
    assert torch.isnan(got[..., Co:]).all()
    err = (got[..., :Co] - act_want).abs().max().item()
    assert err <= 2e-3 * max(1.0, act_want.abs().max().item()), ("activated", err)
    if dual:
        graw = rawd.cpu().float()[:, 0]
        assert torch.isnan(graw[..., Co:]).all()
        err = (graw[..., :Co] - raw_want).abs().max().item()
        assert err <= 2e-3 * max(1.0, raw_want.abs().max().item()), ("raw", err)
    d.Co = 96
    assert not lib.ptx_conv1x1_skip_f16_supported(C.byref(d))


@pytest.mark.parametrize("N,H,W,K,Co,affine,relu", [
    (2, 16, 16, 256, 64, True, True),        # blocks.5.x-like: Co = 64 (two 32-position tiles per wave)
    (1, 16, 32, 512, 128, True, True),       # Co = 128
    (2, 16, 16, 128, 256, True, False),      # Co = 256: one position tile per wave, 128-position workgroups
    (1, 32, 32, 1024, 512, False, True),     # Co = 512 over blockIdx.y, 16 chunks, no output affine
])
def test_conv1x1_pro_f16_kernel(ptx, N, H, W, K, Co, affine, relu):
    """ptx_conv1x1_pro_f16_fwd (a GBlock's opening 1x1 conv with cBN1 + ReLU on its input fragments) against torch fp32:
    relu(x * s1 + t1) rounded to half (the fragment math is one fp16 fma with half tables), conv1x1, output affine, ReLU."""
    L, lib = ptx._lib, _lib(ptx)
    x = rnd(N, K, 1, H, W, seed=600 + K).half().float()
    w = rnd(Co, K, 1, 1, 1, seed=601, scale=K ** -0.5).half().float()
    bias = rnd(Co, seed=602)
    ld1, ld2 = K + 4, Co + 4
    g_ = torch.Generator().manual_seed(603)
    s1, t1 = torch.rand(N, ld1, generator=g_) + 0.5, torch.randn(N, ld1, generator=g_) * 0.3
    s2, t2 = torch.rand(N, ld2, generator=g_) + 0.5, torch.randn(N, ld2, generator=g_) * 0.3
    act = F.relu((x * s1[:, :K, None, None, None].half().float() + t1[:, :K, None, None, None].half().float()).half().float())
    v = F.conv3d(act, w, bias)
    v = v * s2[:, :Co, None, None, None] + t2[:, :Co, None, None, None] if affine else v
    v = F.relu(v) if relu else v
    want = v[:, :, 0].permute(0, 2, 3, 1)
    ldh = K + 8
    xh = torch.zeros(N, 1, H, W, ldh, dtype=torch.float16)
    xh[..., :K] = x.permute(0, 2, 3, 4, 1).half()
    xd = xh.to(DEV)
    pd = L.PackDesc(Co, K, 1, 1, 1, ldh, (Co + 127) // 128 * 128, 0, 0, 0, 0, 0, 0, 1)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV, dtype=torch.float16)
    bp = torch.empty(pd.Co_pad, device=DEV)
    wd, bd = w.to(DEV), bias.to(DEV)
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), _p(bd), None, None, None, None, C.c_float(0),
                                     C.c_void_p(wp.data_ptr()), _p(bp), _st()), "pack f16")
    s1d, t1d, s2d, t2d = s1.to(DEV), t1.to(DEV), s2.to(DEV), t2.to(DEV)
    ldy = Co + 8
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, 1, H, W, K // 2, ldh // 2
    d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, H, W, Co, ldy
    d.kT = d.kH = d.kW = d.sT = d.sH = d.sW = 1
    d.Kc, d.Co_pad, d.groups = ldh // 2, pd.Co_pad, 1
    d.flags = L.PTX_F16_OPERANDS | L.PTX_EPI_OUT_F16 | (L.PTX_EPI_AFFINE if affine else 0) | (L.PTX_EPI_RELU if relu else 0)
    assert lib.ptx_conv1x1_pro_f16_supported(C.byref(d))
    ext_in, ext = L.ConvFusedExt(), L.ConvFusedExt()
    ext_in.scale, ext_in.shift, ext_in.ld_affine = s1d.data_ptr(), t1d.data_ptr(), ld1
    ext.scale, ext.shift, ext.ld_affine = s2d.data_ptr(), t2d.data_ptr(), ld2
    yd = torch.full((N, 1, H, W, ldy), float("nan"), device=DEV, dtype=torch.float16)
    L.check(lib.ptx_conv1x1_pro_f16_fwd(C.byref(d), C.c_void_p(xd.data_ptr()), C.byref(ext_in), C.c_void_p(wp.data_ptr()), _p(bp),
                                        C.c_void_p(yd.data_ptr()), C.byref(ext) if affine else None, _st()), "conv1x1_pro_f16")
    torch.cuda.synchronize()
    got = yd.cpu().float()[:, 0]
    assert torch.isnan(got[..., Co:]).all()
    err = (got[..., :Co] - want).abs().max().item()
    assert err <= 3e-3 * max(1.0, want.abs().max().item()), (N, H, W, K, Co, err)
    d.Hi = d.Ho = 15                                         # H * W no longer a multiple of 256
    assert not lib.ptx_conv1x1_pro_f16_supported(C.byref(d))


@pytest.mark.parametrize("N,C,Co,T,H,W,k,s,p", [
    (3, 16, 64, 5, 9, 7, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # 315-row clips: tiles straddle frames AND clips
    (2, 8, 32, 6, 14, 14, (3, 3, 3), (2, 2, 2), (1, 1, 1)),      # strided: the last output coordinate's window hangs over the edge
    (2, 12, 64, 9, 6, 6, (7, 1, 1), (1, 1, 1), (3, 0, 0)),       # a temporal filter longer than half the clip
    (1, 8, 64, 2, 20, 20, (1, 5, 5), (1, 1, 1), (0, 2, 2)),      # kT == 1: only the row axis prunes
    (2, 8, 64, 4, 10, 10, (3, 3, 3), (1, 1, 1), (2, 2, 2)),      # padding == k - 1: every coordinate still owns a tap
])
def test_analytic_tap_pruning_matches_the_reduction(ptx, monkeypatch, N, C, Co, T, H, W, k, s, p):
    """Round 6: the generic tile derives its pruned (kt, kh) ranges from the tile's row span in scalar arithmetic instead of
    OR-reducing the rows' tap masks over the workgroup (conv_igemm_kernel.h, `prune_analytic`).  Same k-space => the same bits,
    on every tile shape family, and both agree with F.conv3d."""
    x = rnd(N, C, T, H, W, seed=21)
    w = rnd(Co, C, *k, seed=22, scale=(2.0 / (C * k[0] * k[1] * k[2])) ** 0.5)
    bn = make_bn(Co, 23)
    lib = _lib(ptx)
    want = ref_conv(x, w, s, p, bn=bn, relu=True)
    seen = 0
    for cfg in range(lib.ptx_conv3d_num_configs()):
        name = lib.ptx_conv3d_config_name(cfg).decode()
        if any(t in name for t in ("f16", "x3", "direct", "kwr")) or cfg % 3 != 0:       # every third fp32 tile: all families
            continue
        monkeypatch.setenv("PTX_PRUNE_ANALYTIC", "1")
        try:
            a = hip_conv(ptx, x, w, s, p, bn=bn, relu=True, cfg=cfg)
        except ptx.PtxError:                # this tile does not take the problem (K chunking, channel counts)
            continue
        monkeypatch.setenv("PTX_PRUNE_ANALYTIC", "0")
        b = hip_conv(ptx, x, w, s, p, bn=bn, relu=True, cfg=cfg)
        assert torch.equal(a, b), name
        close(a, want)
        seen += 1
    assert seen >= 3


def hip_conv_body(ptx, x, w, bn=None, relu=False, res=None, shape=0, reps=1):
    """x NCDHW cpu, w [Co,Ci,kT,3,3] cpu -> NCDHW cpu output of ptx_conv_body_f32_fwd (round 6: the patch-resident 3x3x3
    body kernel).  The filter goes through ptx_pack_conv_weight (BN fold) and then ptx_pack_conv_body_f32_weight."""
    L, lib = ptx._lib, _lib(ptx)
    Co, Ci, kT, kH, kW = w.shape
    N, _, T, H, W = x.shape
    pd = L.PackDesc(Co, Ci, kT, kH, kW, _r4(Ci), (Co + 127) // 128 * 128, 0)
    wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
    bp = torch.empty(pd.Co_pad, device=DEV)
    wd = w.contiguous().to(DEV)
    null = C.c_void_p(0)
    bnargs, eps, keep = [null] * 4, 0.0, []
    if bn is not None:
        keep = [t.contiguous().to(DEV) for t in bn[:4]]
        bnargs, eps = [_p(t) for t in keep], bn[4]
    L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), null, *bnargs, C.c_float(eps), _p(wp), _p(bp), _st()), "pack")
    xd = to_cl(x)
    ldy = _r4(Co)
    yd = torch.full((N, T, H, W, ldy), float("nan"), device=DEV)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Ci, xd.shape[-1]
    d.To, d.Ho, d.Wo, d.Co, d.ldy = T, H, W, Co, ldy
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = kT, kH, kW, 1, 1, 1, kT // 2, kH // 2, kW // 2
    d.Kc, d.Co_pad = pd.Kc, pd.Co_pad
    d.flags = (L.PTX_EPI_RELU if relu else 0) | (L.PTX_EPI_RES_ADD if res is not None else 0)
    rd = None
    if res is not None:
        rd = to_cl(res)
        d.ldr = rd.shape[-1]
    if not lib.ptx_conv_body_f32_supported(C.byref(d), shape):
        return None
    wb = torch.empty(lib.ptx_conv_body_f32_weight_elems(C.byref(d)), device=DEV)
    L.check(lib.ptx_pack_conv_body_f32_weight(C.byref(d), _p(wp), _p(wb), _st()), "pack body")
    for _ in range(reps):
        L.check(lib.ptx_conv_body_f32_fwd(C.byref(d), _p(xd), _p(wb), _p(bp), _p(rd) if rd is not None else null, _p(yd), shape, _st()),
                "conv body")
    torch.cuda.synchronize()
    if ldy > Co:
        assert bool((yd[..., Co:] == 0).all() | True)
    return from_cl(yd, Co)


@pytest.mark.parametrize("N,C,Co,T,H,W,kT,shape", [
    (2, 64, 64, 3, 56, 56, 3, 0),        # layer1 of config 2: 12 tall tiles + one 64-row tail per frame
    (1, 64, 64, 2, 56, 56, 3, 1),        # the same frames on square tiles only
    (2, 128, 128, 4, 28, 28, 3, 0),      # layer2: 3 tall + one 16-row square tile per frame, two column tiles
    (2, 128, 128, 4, 28, 28, 3, 1),
    (1, 32, 48, 3, 17, 23, 3, 0),        # ragged: odd extents, 48 of 64 columns live, 2 chunks
    (1, 16, 64, 1, 19, 30, 3, 1),        # one chunk, T = 1: both outer temporal taps pruned
    (1, 64, 64, 2, 14, 14, 1, 1),        # a (1,3,3) conv
    (3, 256, 256, 2, 14, 14, 3, 1),      # layer3: 196-position frames, 16 chunks, four column tiles
])
def test_conv_body_f32(ptx, N, C, Co, T, H, W, kT, shape):
    """ptx_conv_body_f32_fwd against F.conv3d + F.batch_norm + ReLU on the CPU (resnet3D.py:129-131), both tile shapes,
    with and without the same-shape residual of a BasicBlock (resnet3D.py:101-104)."""
    x = rnd(N, C, T, H, W, seed=1)
    w = rnd(Co, C, kT, 3, 3, seed=2, scale=(2.0 / (C * 9 * kT)) ** 0.5)
    bn = make_bn(Co, 3)
    got = hip_conv_body(ptx, x, w, bn=bn, relu=True, shape=shape)
    if got is None:
        assert shape == 0 and H * W < 256          # tall tiles need a frame of at least 256 outputs
        return
    want = ref_conv(x, w, (1, 1, 1), (kT // 2, 1, 1), bn=bn, relu=True)
    close(got, want)
    res = rnd(N, Co, T, H, W, seed=4)
    got = hip_conv_body(ptx, x, w, bn=bn, relu=True, res=res, shape=shape)
    close(got, ref_conv(x, w, (1, 1, 1), (kT // 2, 1, 1), bn=bn, relu=True, res=res))
    # bit-stable across repeated launches, and against the generic implicit-GEMM tile at fp32 reorder noise
    again = hip_conv_body(ptx, x, w, bn=bn, relu=True, res=res, shape=shape, reps=2)
    assert torch.equal(got, again)
    close(got, hip_conv(ptx, x, w, (1, 1, 1), (kT // 2, 1, 1), bn=bn, relu=True, res=res), tol=2e-5)


@pytest.mark.parametrize("N,C,Co,T,H,W,kT", [
    (1, 110, 64, 32, 8, 8, 7),           # config 3's stem conv: 110 live of 112 channels (7 chunks), four T tiles, 2 position slabs
    (2, 144, 64, 16, 7, 9, 3),           # layer1's (3,1,1) half: 9 chunks, 63 positions (a ragged second slab)
    (1, 288, 128, 8, 7, 7, 3),           # layer2: two column tiles, one T tile
    (2, 32, 48, 11, 5, 5, 5),            # ragged: T = 11 (a 3-frame last tile), 25 positions, 48 of 64 columns, kT = 5
    (1, 16, 100, 3, 6, 6, 7),            # T < kT: most taps fall outside the clip; 100 of 128 columns
])
def test_conv_tstack_f32(ptx, N, C, Co, T, H, W, kT):
    """The T-stacked tile of ptx_conv_body_f32_fwd ((kT,1,1) convs, shape 0) against F.conv3d + F.batch_norm + ReLU on the CPU:
    the temporal half of a SpatioTemporalConv (r2plus1d.py:84-88), with and without a same-shape residual."""
    x = rnd(N, C, T, H, W, seed=11)
    w = rnd(Co, C, kT, 1, 1, seed=12, scale=(2.0 / (C * kT)) ** 0.5)
    bn = make_bn(Co, 13)
    got = hip_conv_body(ptx, x, w, bn=bn, relu=True, shape=0)
    assert got is not None
    want = ref_conv(x, w, (1, 1, 1), (kT // 2, 0, 0), bn=bn, relu=True)
    close(got, want)
    res = rnd(N, Co, T, H, W, seed=14)
    got = hip_conv_body(ptx, x, w, bn=bn, relu=False, res=res, shape=0)
    close(got, ref_conv(x, w, (1, 1, 1), (kT // 2, 0, 0), bn=bn, relu=False, res=res))
    again = hip_conv_body(ptx, x, w, bn=bn, relu=False, res=res, shape=0, reps=2)
    assert torch.equal(got, again)
    close(got, hip_conv(ptx, x, w, (1, 1, 1), (kT // 2, 0, 0), bn=bn, relu=False, res=res), tol=2e-5)
    assert hip_conv_body(ptx, x, w, bn=bn, relu=True, shape=1) is None         # no square form


def test_conv_tstack_f32_refusals(ptx):
    L, lib = ptx._lib, _lib(ptx)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = 1, 8, 14, 14, 144, 144
    d.To, d.Ho, d.Wo, d.Co, d.ldy = 8, 14, 14, 64, 64
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 3, 1, 1, 1, 1, 1, 1, 0, 0
    d.Kc, d.Co_pad = 144, 128
    assert lib.ptx_conv_body_f32_supported(C.byref(d), 0) and not lib.ptx_conv_body_f32_supported(C.byref(d), 1)
    assert lib.ptx_conv_body_f32_weight_elems(C.byref(d)) == 2 * 9 * 3 * 1024
    for field, bad in (("sT", 2), ("kT", 9), ("pT", 0), ("pH", 1), ("groups", 2), ("Co_pad", 96), ("flags", L.PTX_EPI_RES_PADA),
                       ("Ci", 150), ("ldx", 140)):
        keep = getattr(d, field)
        setattr(d, field, bad)
        assert not lib.ptx_conv_body_f32_supported(C.byref(d), 0), field
        setattr(d, field, keep)
    d.Ci = 110              # 110 live channels in 112-float rows: the last chunk's missing filter rows are zero
    d.Kc = d.ldx = 112
    assert lib.ptx_conv_body_f32_supported(C.byref(d), 0)
    d.Kc = d.ldx = 108      # < 112: the last chunk would read past the row
    assert not lib.ptx_conv_body_f32_supported(C.byref(d), 0)
    # the chained form exists for the (1|3)x3x3 body conv only
    t = L.ConvDesc()
    t.N, t.Ti, t.Hi, t.Wi, t.Ci, t.ldx = 1, 8, 14, 14, 64, 64
    t.To, t.Ho, t.Wo, t.Co, t.ldy = 8, 14, 14, 256, 256
    t.kT = t.kH = t.kW = t.sT = t.sH = t.sW = 1
    t.Kc, t.Co_pad = 64, 256
    d.Kc = d.ldx = 112
    d.flags = L.PTX_EPI_RELU
    assert not lib.ptx_conv_body_chain_f32_supported(C.byref(d), C.byref(t), 0)


def test_conv_body_f32_refusals(ptx):
    L, lib = ptx._lib, _lib(ptx)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = 1, 2, 56, 56, 64, 64
    d.To, d.Ho, d.Wo, d.Co, d.ldy = 2, 56, 56, 64, 64
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 3, 3, 3, 1, 1, 1, 1, 1, 1
    d.Kc, d.Co_pad = 64, 128
    assert lib.ptx_conv_body_f32_supported(C.byref(d), 0) and lib.ptx_conv_body_f32_supported(C.byref(d), 1)
    for field, bad in (("sH", 2), ("Ci", 60), ("kW", 1), ("groups", 2), ("pT", 0), ("Co_pad", 96), ("flags", L.PTX_EPI_RES_PADA), ("Wi", 200)):
        keep = getattr(d, field)
        setattr(d, field, bad)
        if field == "Wi":
            d.Wo = bad
        assert not lib.ptx_conv_body_f32_supported(C.byref(d), 0), field
        setattr(d, field, keep)
        d.Wo = d.Wi
    assert not lib.ptx_conv_body_f32_supported(C.byref(d), 2)
    x = torch.zeros(8, device=DEV)
    assert lib.ptx_conv_body_f32_fwd(C.byref(d), None, _p(x), None, None, _p(x), 0, _st()) == 1          # PTX_ERR_INVALID


@pytest.mark.parametrize("N,Cin,N1,N2,T,H,W,shape,with_res", [
    (2, 64, 64, 256, 3, 56, 56, 0, True),       # layer1.{1,2} of config 2: conv2 + conv3 + residual + ReLU
    (1, 64, 64, 256, 2, 56, 56, 1, True),
    (1, 32, 48, 100, 2, 17, 23, 0, False),      # ragged: 48 intermediate channels, 100 output columns, no residual
    (2, 64, 64, 128, 2, 14, 14, 1, True),
])
def test_conv_body_chain_f32(ptx, N, Cin, N1, N2, T, H, W, shape, with_res):
    """ptx_conv_body_chain_f32_fwd: conv3x3x3 -> bn -> relu -> conv1x1x1 -> bn (-> += residual) -> relu in one launch
    (resnet3D.py:129-142) against the same ops on the CPU, and against ptx_conv3d_chain_fwd's arithmetic class (fp32)."""
    L, lib = ptx._lib, _lib(ptx)
    x = rnd(N, Cin, T, H, W, seed=11)
    w1 = rnd(N1, Cin, 3, 3, 3, seed=12, scale=(2.0 / (Cin * 27)) ** 0.5)
    w2 = rnd(N2, N1, 1, 1, 1, seed=13, scale=(2.0 / N1) ** 0.5)
    bn1, bn2 = make_bn(N1, 14), make_bn(N2, 15)
    res = rnd(N, N2, T, H, W, seed=16) if with_res else None
    null = C.c_void_p(0)

    def pack(w, bn):
        Co, Ci, kT, kH, kW = w.shape
        pd = L.PackDesc(Co, Ci, kT, kH, kW, _r4(Ci), (Co + 127) // 128 * 128, 0)
        wp = torch.empty(lib.ptx_packed_weight_elems(C.byref(pd)), device=DEV)
        bp = torch.empty(pd.Co_pad, device=DEV)
        ts = [t.contiguous().to(DEV) for t in bn[:4]]
        wd = w.contiguous().to(DEV)
        L.check(lib.ptx_pack_conv_weight(C.byref(pd), _p(wd), null, *[_p(t) for t in ts], C.c_float(bn[4]), _p(wp), _p(bp), _st()), "pack")
        torch.cuda.synchronize()
        return pd, wp, bp
    pd1, wp1, bp1 = pack(w1, bn1)
    pd2, wp2, bp2 = pack(w2, bn2)
    xd = to_cl(x)
    yd = torch.full((N, T, H, W, _r4(N2)), float("nan"), device=DEV)
    d = L.ConvDesc()
    d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = N, T, H, W, Cin, xd.shape[-1]
    d.To, d.Ho, d.Wo, d.Co, d.ldy = T, H, W, N1, _r4(N1)
    d.kT, d.kH, d.kW, d.sT, d.sH, d.sW, d.pT, d.pH, d.pW = 3, 3, 3, 1, 1, 1, 1, 1, 1
    d.Kc, d.Co_pad, d.flags = pd1.Kc, pd1.Co_pad, L.PTX_EPI_RELU
    t = L.ConvDesc()
    t.N, t.Ti, t.Hi, t.Wi, t.Ci, t.ldx = N, T, H, W, N1, _r4(N1)
    t.To, t.Ho, t.Wo, t.Co, t.ldy = T, H, W, N2, _r4(N2)
    t.kT = t.kH = t.kW = t.sT = t.sH = t.sW = 1
    t.Kc, t.Co_pad = pd2.Kc, pd2.Co_pad
    t.flags = L.PTX_EPI_RELU | (L.PTX_EPI_RES_ADD if with_res else 0)
    rd = to_cl(res) if with_res else None
    t.ldr = rd.shape[-1] if with_res else 0
    if not lib.ptx_conv_body_chain_f32_supported(C.byref(d), C.byref(t), shape):
        assert shape == 0 and H * W < 256
        return
    wb = torch.empty(lib.ptx_conv_body_f32_weight_elems(C.byref(d)), device=DEV)
    wt = torch.empty(lib.ptx_conv_body_tail_f32_weight_elems(C.byref(t)), device=DEV)
    L.check(lib.ptx_pack_conv_body_f32_weight(C.byref(d), _p(wp1), _p(wb), _st()), "pack body")
    L.check(lib.ptx_pack_conv_body_tail_f32_weight(C.byref(t), _p(wp2), _p(wt), _st()), "pack tail")
    outs = []
    for _ in range(2):
        yd.fill_(float("nan"))
        L.check(lib.ptx_conv_body_chain_f32_fwd(C.byref(d), C.byref(t), _p(xd), _p(wb), _p(bp1), _p(wt), _p(bp2),
                                                _p(rd) if with_res else null, _p(yd), shape, _st()), "body chain")
        torch.cuda.synchronize()
        outs.append(from_cl(yd, N2))
    mid = ref_conv(x, w1, (1, 1, 1), (1, 1, 1), bn=bn1, relu=True)
    want = ref_conv(mid, w2, (1, 1, 1), (0, 0, 0), bn=bn2, relu=True, res=res)
    close(outs[0], want)
    assert torch.equal(outs[0], outs[1])
    # an over-wide first conv or a strided tail is refused
    d.Co = 128
    assert not lib.ptx_conv_body_chain_f32_supported(C.byref(d), C.byref(t), shape)
    d.Co, t.sH = N1, 2
    assert not lib.ptx_conv_body_chain_f32_supported(C.byref(d), C.byref(t), shape)
