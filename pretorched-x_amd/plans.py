"""Plan builders of the model families beyond the ResNet trunk: SlowFast, I3D, BigGAN-deep.

Each `build_*(plan, model)` appends launches to an `engine.Plan` (named `self` below: these were Plan
methods and read like them) using its building blocks -- conv / conv_bn / maxpool / act / slices --
and sets `plan.feat` (and `plan.head` when the classifier tail is not the default avg-pool + Linear).
"""
import ctypes as C

import torch
import torch.nn as nn

from ._lib import PTX_ACT_OUT_F16, PtxError, check
from .engine import RawInput, _geom, _ptr, _r4, _stream


# --------------------------------------------------------------------------------------------
# SlowFast (slowfast.py:102-398)
def build_slowfast(self, model):
    N, Cin, T, H, W = self.shape
    arch, mode = model.arch, model.mode
    pool = ((1, 3, 3), (1, 2, 2), (0, 1, 1))
    stages = ("res2", "res3", "res4", "res5")

    def out_hw(blk, x):
        # (1,3,3) pad (0,1,1) conv carrying the block's spatial stride (slowfast.py:16-20,72-74)
        return (x.H - 1) // blk.stride + 1, (x.W - 1) // blk.stride + 1

    fast_feat = slow_feat = None
    taps = []                                            # fast activations feeding the lateral convs
    if mode in ("sf", "f"):
        fast = model.fast if mode == "sf" else model
        step = model.fast_stride
        raw = RawInput(N, Cin, len(range(0, T, step)), H, W, t_step=step, T_full=T, norm=self.norm)
        f = self.conv_bn(raw, fast.conv1, fast.bn1, relu=True, label="fast.conv1")
        f = self.maxpool(f, *pool)
        taps.append(f)
        for name in stages:
            for bi, blk in enumerate(getattr(fast, name)):
                f = self._block(arch, blk, f, "fast.%s.%d" % (name, bi))
            taps.append(f)
        fast_feat = f
    if mode in ("sf", "s"):
        slow = model.slow if mode == "sf" else model
        step = model.slow_stride
        raw = RawInput(N, Cin, len(range(0, T, step)), H, W, t_step=step, T_full=T, norm=self.norm)
        s = self.conv_bn(raw, slow.conv1, slow.bn1, relu=True, label="slow.conv1")
        laterals = ([model.fast.lateral_p1, model.fast.lateral_res2, model.fast.lateral_res3,
                     model.fast.lateral_res4] if mode == "sf" else None)

        def fuse_lateral(i, main_C, dims):
            """Allocate cat([main, lateral_i(fast tap i)], dim=1) and emit the lateral conv into its slice."""
            lat = laterals[i]
            k, st_, pd = _geom(lat)
            src = taps[i]
            lt = ((src.T + 2 * pd[0] - k[0]) // st_[0] + 1, src.H, src.W)
            if lt != dims:
                raise PtxError("SlowFast: lateral %d yields %s but the slow pathway is %s at that stage -- "
                               "torch.cat would fail in the reference too (slowfast.py:145-151); the lateral convs "
                               "stride time by 8, so slow_stride must be 8 * fast_stride" % (i, lt, dims))
            cat = self.act(N, dims[0], dims[1], dims[2], main_C + lat.out_channels)
            self.conv(src, self.pack(lat, None), st_, pd, y=cat.slice(main_C, lat.out_channels),
                      label="fast.lateral%d" % i)
            return cat

        if laterals:
            Ho, Wo = (s.H - 1) // 2 + 1, (s.W - 1) // 2 + 1
            x = fuse_lateral(0, s.C, (s.T, Ho, Wo))
            self.maxpool(s, *pool, y=x.slice(0, s.C))
        else:
            x = self.maxpool(s, *pool)
        for si, name in enumerate(stages):
            blocks = list(getattr(slow, name))
            for bi, blk in enumerate(blocks):
                label = "slow.%s.%d" % (name, bi)
                if laterals and bi == len(blocks) - 1 and si < 3:
                    Ho, Wo = out_hw(blk, x)
                    cout = blk.out_channels
                    nxt = fuse_lateral(si + 1, cout, (x.T, Ho, Wo))
                    self._block(arch, blk, x, label, out=nxt.slice(0, cout))
                    x = nxt
                else:
                    x = self._block(arch, blk, x, label)
        slow_feat = x
    if mode != "sf":
        self.feat = slow_feat if mode == "s" else fast_feat
        self.pooled = torch.empty((N, self.feat.C), device=self.dev, dtype=torch.float32)
        return
    # two-pathway head: cat([avgpool(slow), avgpool(fast)]) -> dropout (identity) -> last_linear
    self.feat = slow_feat
    cs, cf = slow_feat.C, fast_feat.C
    ps = torch.empty((N, cs), device=self.dev, dtype=torch.float32)
    pf = torch.empty((N, cf), device=self.dev, dtype=torch.float32)
    self.pooled = torch.empty((N, cs + cf), device=self.dev, dtype=torch.float32)
    lib, pooled = self.lib, self.pooled

    def head(engine, model):
        st = _stream()
        for f_, p_ in ((slow_feat, ps), (fast_feat, pf)):
            check(lib.ptx_global_avgpool(_ptr(f_.t), _ptr(p_), f_.N, f_.C, f_.S, f_.ld, 0, st), "ptx_global_avgpool")
        check(lib.ptx_copy2d(_ptr(ps), _ptr(pooled), N, cs, cs, cs + cf, st), "ptx_copy2d")
        check(lib.ptx_copy2d(_ptr(pf), _ptr(pooled, cs), N, cf, cf, cs + cf, st), "ptx_copy2d")
        out = engine._head(model, _ptr(pooled), N, cs + cf, self.dev)
        return out if out is not None else model.head_module(pooled.clone())
    self.head = head

# --------------------------------------------------------------------------------------------
# I3D (Inception-v1 3-D)
def build_i3d(self, model):
    N, Cin, T, H, W = self.shape
    one = (1, 1, 1)

    def unit(x, u, label, y=None):
        conv = u.conv3d
        k, s, _ = _geom(conv)
        bn = u.bn if u.has_bn else None
        if isinstance(x, RawInput):
            return self.conv_bn(x, conv, bn, relu=True, label=label)
        return self.conv(x, self.pack(conv, bn), s, (0, 0, 0), relu=u.has_bn, label=label, y=y, same=True)

    raw = RawInput(N, Cin, T, H, W, norm=self.norm)
    x = unit(raw, model.Conv3d_1a_7x7, "Conv3d_1a_7x7")
    x = self.maxpool(x, (1, 3, 3), (1, 2, 2), None, same=True)
    x = unit(x, model.Conv3d_2b_1x1, "Conv3d_2b_1x1")
    x = unit(x, model.Conv3d_2c_3x3, "Conv3d_2c_3x3")
    x = self.maxpool(x, (1, 3, 3), (1, 2, 2), None, same=True)
    for entry in model.layout:
        name = entry[0]
        if name.startswith("pool"):
            x = self.maxpool(x, entry[1], entry[2], None, same=True)
            continue
        m = getattr(model, name)
        out = self.act(x.N, x.T, x.H, x.W, m.out_channels)
        c1, c2, c3 = m.splits[0], m.splits[0] + m.splits[1], m.splits[0] + m.splits[1] + m.splits[2]
        # the four branches write their channel slices of the module output: no torch.cat
        unit(x, m.b0, name + ".b0", y=out.slice(0, m.splits[0]))
        unit(unit(x, m.b1a, name + ".b1a"), m.b1b, name + ".b1b", y=out.slice(c1, m.splits[1]))
        unit(unit(x, m.b2a, name + ".b2a"), m.b2b, name + ".b2b", y=out.slice(c2, m.splits[2]))
        pooled = self.maxpool(x, (3, 3, 3), one, None, same=True)
        unit(pooled, m.b3b, name + ".b3b", y=out.slice(c3, m.splits[3]))
        x = out
    self.feat = x
    # head: avg_pool3d([2,7,7], stride 1) -> dropout (identity) -> 1x1x1 conv with bias -> squeeze ->
    # mean over the remaining time steps
    if (x.H, x.W) != (7, 7) or x.T < 2:
        self.head = None
        self.pooled = None
        self.head_error = ("I3D head: AvgPool3d([2,7,7]) needs a [T>=2,7,7] Mixed_5c map (224x224 input, >=16 "
                           "frames); got [%d,%d,%d]" % (x.T, x.H, x.W))
        return
    Tf, Cf = x.T, x.C
    frame_mean = torch.empty((N * Tf, Cf), device=self.dev, dtype=torch.float32)
    win = torch.empty((N * (Tf - 1), Cf), device=self.dev, dtype=torch.float32)
    self.pooled = win
    lib = self.lib

    def head(engine, model):
        st = _stream()
        conv = model.head_module
        if not isinstance(conv, nn.Conv3d):
            raise PtxError("I3D: model.logits.conv3d must be a 1x1x1 Conv3d")
        ncls = conv.out_channels
        check(lib.ptx_global_avgpool(_ptr(x.t), _ptr(frame_mean), N * Tf, Cf, x.H * x.W, x.ld, 0, st),
              "ptx_global_avgpool")
        check(lib.ptx_window_mean(_ptr(frame_mean), _ptr(win), N, Tf, Cf, 2, 1, st), "ptx_window_mean")
        w = conv.weight.detach().reshape(ncls, Cf).contiguous()
        b = conv.bias.detach().contiguous() if conv.bias is not None else None
        per_frame = torch.empty((N * (Tf - 1), ncls), device=self.dev, dtype=torch.float32)
        check(lib.ptx_linear_fwd(_ptr(win), _ptr(w), _ptr(b) if b is not None else C.c_void_p(0), _ptr(per_frame),
                                 N * (Tf - 1), Cf, ncls, Cf, ncls, 0, st), "ptx_linear_fwd")
        out = torch.empty((N, ncls), device=self.dev, dtype=torch.float32)
        check(lib.ptx_window_mean(_ptr(per_frame), _ptr(out), N, Tf - 1, ncls, Tf - 1, 1, st), "ptx_window_mean")
        return out
    self.head = head

# --------------------------------------------------------------------------------------------
# MNISTNonLocalNet (nonlocalnet.py:273-309)
def build_mnist_nl(self, model):
    N, Cin, H, W = self.shape
    x = self.to_channels_last(RawInput(N, Cin, 1, H, W))
    for i in (0, 5, 10):                          # conv3x3 (bias) + BN + ReLU as one launch, MaxPool2d(2), NL block
        x = self.conv_bn(x, model.convs[i], model.convs[i + 1], relu=True, label="convs.%d" % i)
        x = self.maxpool(x, (1, 2, 2), (1, 2, 2), (0, 0, 0))
        if i < 10:
            x = self.nonlocal_block(x, model.convs[i + 4], "convs.%d" % (i + 4))
    self.feat = x
    self.pooled = None
    if x.C * x.H * x.W != model.fc[0].in_features:
        self.head = None
        self.head_error = ("MNISTNonLocalNet: fc expects %d features (a 28x28 input), the conv stack produced %dx%dx%d" % (
            model.fc[0].in_features, x.C, x.H, x.W))
        return
    flat = torch.empty((N, x.C, x.H, x.W), device=self.dev, dtype=torch.float32)      # NCHW: `.view(batch, -1)` order
    self.keepalive.append(flat)
    lib = self.lib

    def head(engine, model, x=x, flat=flat):
        from ._lib import PTX_EPI_RELU
        from .engine import linear
        check(lib.ptx_ndhwc_to_ncdhw(_ptr(x.t), _ptr(flat), x.N, x.C, x.S, x.ld, _stream()), "ptx_ndhwc_to_ncdhw")
        h = linear(flat.view(x.N, -1), model.fc[0], PTX_EPI_RELU)          # Dropout is the identity in eval mode
        return linear(h, model.fc[3])
    self.head = head


# --------------------------------------------------------------------------------------------
# BigGAN-deep generator
def build_biggan(self, model):
    N = self.shape[0]
    lib, dev = self.lib, self.dev
    cond_dim, sdim, eps = model.cond_dim, model.shared_dim, float(model.bn_eps)
    f32 = dict(device=dev, dtype=torch.float32)
    cond = torch.empty((N, cond_dim), **f32)
    # every conditional BN of the network, in execution order -> one [sum C] table
    ccbns = []
    for stage in model.blocks:
        for blk in stage:
            if blk.kind == "gblock":
                ccbns += [blk.bn1, blk.bn2, blk.bn3, blk.bn4]
    offs, tot = {}, 0
    for bn in ccbns:
        offs[id(bn)] = tot
        tot += bn.channels
    wg, wb = torch.empty((tot, cond_dim), **f32), torch.empty((tot, cond_dim), **f32)
    mean_all, var_all = torch.empty(tot, **f32), torch.empty(tot, **f32)
    gain_all, bias_all = torch.empty((N, tot), **f32), torch.empty((N, tot), **f32)
    scale_all, shift_all = torch.empty((N, tot), **f32), torch.empty((N, tot), **f32)
    obn = model.output_layer[0]
    oscale, oshift = torch.empty((N, obn.channels), **f32), torch.empty((N, obn.channels), **f32)
    bw = model.bottom_width
    c0 = model.linear.out_features // (bw * bw)
    w0, b0 = torch.empty((bw * bw * c0, cond_dim), **f32), torch.empty(bw * bw * c0, **f32)
    self.keepalive += [cond, wg, wb, mean_all, var_all, gain_all, bias_all, scale_all, shift_all, oscale, oshift, w0, b0]

    bn_refs = [(self.ref(bn), offs[id(bn)], bn.channels) for bn in ccbns]
    lin_ref, obn_ref = self.ref(model.linear), self.ref(obn)

    def refresh_tables():
        # weight relayout only (concatenation / row permutation), rebuilt when a parameter changes
        linear = self.get(lin_ref)
        for r, o, c in bn_refs:
            bn = self.get(r)
            wg[o:o + c].copy_(bn.gain.weight.detach())
            wb[o:o + c].copy_(bn.bias.weight.detach())
            mean_all[o:o + c].copy_(bn.stored_mean)
            var_all[o:o + c].copy_(bn.stored_var)
        # first Linear emits NCHW-ordered features (c, h, w); permute its rows so it writes NHWC directly
        w0.copy_(linear.weight.detach().view(c0, bw, bw, cond_dim).permute(1, 2, 0, 3).reshape(-1, cond_dim))
        b0.copy_(linear.bias.detach().view(c0, bw, bw).permute(1, 2, 0).reshape(-1))
    if torch.device(dev).type != "meta":
        self.refreshers.append(refresh_tables)

    def prologue(st, self=self):
        obn = self.get(obn_ref)
        # y = cat([shared(labels), z], 1); all cBN gains/biases in two GEMVs; fold with the stored statistics
        check(lib.ptx_copy2d(self.in_ptr2, _ptr(cond), N, sdim, sdim, cond_dim, st), "cond.y")
        check(lib.ptx_copy2d(self.in_ptr, _ptr(cond, sdim), N, cond_dim - sdim, cond_dim - sdim, cond_dim, st), "cond.z")
        check(lib.ptx_linear_fwd(_ptr(cond), _ptr(wg), None, _ptr(gain_all), N, cond_dim, tot, cond_dim, tot, 0, st), "cbn.gain")
        check(lib.ptx_linear_fwd(_ptr(cond), _ptr(wb), None, _ptr(bias_all), N, cond_dim, tot, cond_dim, tot, 0, st), "cbn.bias")
        check(lib.ptx_cbn_fold(_ptr(gain_all), _ptr(bias_all), _ptr(mean_all), _ptr(var_all), C.c_float(eps),
                               _ptr(scale_all), _ptr(shift_all), N, tot, tot, tot, tot, 1, st), "cbn.fold")
        check(lib.ptx_cbn_fold(_ptr(obn.gain.detach()), _ptr(obn.bias.detach()), _ptr(obn.stored_mean),
                               _ptr(obn.stored_var), C.c_float(eps), _ptr(oscale), _ptr(oshift), N, obn.channels, 0, 0,
                               obn.channels, 0, st), "bn.fold")
    self.steps.append(prologue)

    h = self.act(N, 1, bw, bw, c0)

    def first_linear(st, hp=_ptr(h.t)):
        check(lib.ptx_linear_fwd(_ptr(cond), _ptr(w0), _ptr(b0), hp, N, cond_dim, bw * bw * c0, cond_dim, bw * bw * c0,
                                 0, st), "linear")
    self.steps.append(first_linear)

    half = getattr(model, "precision", "fp32") == "fp16"      # fp16 operands for every conv behind a cBN pass
    self.half_plan = half

    def affine(x, sc, sh, ld_s, up, act=1, f16_out=None):
        f16_out = (half and act == 1) if f16_out is None else f16_out
        y = self.act(N, 1, x.H * up, x.W * up, x.C, f16=f16_out)
        xp, yp, H_, W_, C_, ldx, ldy = _ptr(x.t), C.c_void_p(y.t.data_ptr()), x.H, x.W, x.C, x.ld, y.ld
        if f16_out:
            act |= PTX_ACT_OUT_F16

        def step(st):
            check(lib.ptx_affine_act_upsample(xp, yp, sc, sh, ld_s, N, H_, W_, C_, ldx, ldy, up, act, st),
                  "ptx_affine_act_upsample")
        from .engine import _tag
        self.steps.append(_tag(step, "affine_act_upsample", 4 * N * H_ * W_ * C_ + (2 if f16_out else 4) * N * H_ * W_ * C_ * up * up))
        return y

    def cbn(x, bn, up=1):
        o = offs[id(bn)]
        return affine(x, _ptr(scale_all, o), _ptr(shift_all, o), tot, up)

    one, zero = (1, 1, 1), (0, 0, 0)
    if half:
        return _biggan_fp16_stages(self, model, h, cbn, affine, offs, scale_all, shift_all, tot, oscale, oshift)
    for si, stage in enumerate(model.blocks):
        for bi, blk in enumerate(stage):
            name = "blocks.%d.%d" % (si, bi)
            if blk.kind == "gblock":
                up = 2 if blk.upsample else 1
                t = self.conv(cbn(h, blk.bn1), self.pack(blk.conv1, None), one, zero, label=name + ".conv1")
                t = self.conv(cbn(t, blk.bn2, up), self.pack(blk.conv2, None), one, (0, 1, 1), label=name + ".conv2")
                t = self.conv(cbn(t, blk.bn3), self.pack(blk.conv3, None), one, (0, 1, 1), label=name + ".conv3")
                if up == 1 and blk.in_channels == blk.out_channels:
                    h = self.conv(cbn(t, blk.bn4), self.pack(blk.conv4, None), one, zero, res=h, label=name + ".conv4")
                else:   # skip = upsample(x[:, :Cout]) gathered in the epilogue
                    h = self.conv(cbn(t, blk.bn4), self.pack(blk.conv4, None), one, zero, res=h, res_kind="up",
                                  res_stride=(0, up // 2, up // 2), label=name + ".conv4")
            else:
                h = biggan_attention(self, h, blk, name)
    a = affine(h, _ptr(oscale), _ptr(oshift), obn.channels, 1)
    img = self.conv(a, self.pack(model.output_layer[2], None), one, (0, 1, 1), label="output_layer.2")
    self.feat = affine(img, None, None, 0, 1, act=2)            # tanh
    self.pooled = None


# ---- guard of the packed-fp16 affine -------------------------------------------------------------
# Two consumer kernels of the fp16 generator plan apply a BatchNorm's folded scale / shift to their INPUT fragments as packed
# fp16 FMAs (ptx_conv1x1_pro_f16_fwd: a GBlock's cBN1 + ReLU; ptx_rgb_conv3x3_f16_fwd: the output layer's BN + ReLU), with the
# tables rounded to halfs.  That is only as good as the tables' range and conditioning:
#   * a table entry above the half range (65504) becomes inf;
#   * v = x * scale + shift with shift = bias - mean * scale cancels when |mean| >> sigma: the fp16 FMA's rounding error is
#     2^-11 |scale * mean| against a signal of |scale| * sigma, i.e. 2^-11 |mean| / sigma of a standard deviation.
# The producer stores a RAW-only map when its consumer takes this path, so the decision is made where the plan is compiled,
# from the model's own parameters: worst-case table magnitudes below HALF_TABLE_MAX for any conditioning vector with
# |cond| <= cond_max, and |mean| / sigma <= HALF_CANCEL_MAX (error <= 0.8 % of a standard deviation per normalisation).
# Otherwise the block keeps the two-output flow, whose affine runs in fp32 in the PRODUCER's epilogue.
HALF_TABLE_MAX = 6.0e4
HALF_CANCEL_MAX = 16.0
COND_ABS_MAX = 6.0           # |z| of a (truncated) normal sample; the class embedding's own maximum is added by the caller


def half_affine_ok(bn, eps, cond_max=COND_ABS_MAX):
    """(ok, (max |scale| bound, max |shift| bound, max |mean| / sigma)) for a (conditional) BatchNorm whose folded tables a
    consumer kernel would apply as packed fp16 FMAs.  Conditional (`gain` / `bias` are bias-free Linears of the conditioning
    vector, gain = 1 + W_g cond): bounds over every |cond|_inf <= cond_max; plain (`gain` / `bias` parameters): exact."""
    with torch.no_grad():
        mean, var = bn.stored_mean.detach().float(), bn.stored_var.detach().float()
        sigma = (var + eps).sqrt()
        if isinstance(bn.gain, torch.nn.Linear):
            g = 1.0 + bn.gain.weight.detach().float().abs().sum(1) * cond_max
            b = bn.bias.weight.detach().float().abs().sum(1) * cond_max
        else:
            g, b = bn.gain.detach().float().abs(), bn.bias.detach().float().abs()
        scale = g / sigma
        shift = b + mean.abs() * scale
        vals = [float(v) for v in torch.stack([scale.max(), shift.max(), (mean.abs() / sigma).max()]).tolist()]
    ok = vals[0] < HALF_TABLE_MAX and vals[1] < HALF_TABLE_MAX and vals[2] <= HALF_CANCEL_MAX
    return bool(ok), tuple(vals)


def _biggan_fp16_stages(self, model, h, cbn, affine, offs, scale_all, shift_all, tot, oscale, oshift):
    """precision='fp16' (BASELINE.json config 5 "fp16 MFMA"): the fused cBN + upsample + conv generator stage.

    Between the convs of a GBlock the activations are halfs and no normalisation / ReLU / upsampling pass exists:
      * the class-conditional BN + ReLU that FOLLOWS a conv runs in that conv's epilogue as a per-sample affine
        (scale / shift tables folded from the one cBN GEMV pair) and the result is stored as halfs;
      * the nearest 2x upsample that PRECEDES the first 3x3 conv of an upsampling block is the conv's loader
        (PTX_PRO_UP2: address arithmetic in the implicit GEMM, the upsampled tensor never exists);
      * a block's last conv adds the skip up(x[:, :Cout]) in its epilogue and writes BOTH what the next block
        consumes: relu(cBN1_next(out)) for its first conv and the raw sum for its skip connection, as halfs;
      * the image conv applies tanh in its epilogue (16-wide N tile on 16x16x32 MFMA for its 3 channels).
    HBM passes left: the first cBN of the network (4x4 input of the linear layer), the pass around the 64x64
    self-attention block (fp32, on the non-local kernels), the NHWC -> NCHW edge of the 3-channel image."""
    one, zero = (1, 1, 1), (0, 0, 0)
    flat = [(si, bi, blk) for si, stage in enumerate(model.blocks) for bi, blk in enumerate(stage)]
    obn = model.output_layer[0]

    def tab(bn):
        o = offs[id(bn)]
        return (_ptr(scale_all, o), _ptr(shift_all, o), tot)

    import os
    eps = float(model.bn_eps)
    with torch.no_grad():
        cond_max = max(COND_ABS_MAX, float(model.shared.weight.detach().abs().max()))
    guard_on = os.environ.get("PTX_HALF_AFFINE_GUARD", "1") != "0"       # 0: A/B runs and the test that shows the failure
    guarded = []                      # (module ref, verdict the plan was compiled with): re-checked when the weights change

    def half_ok(bn):
        ok, _ = half_affine_ok(bn, eps, cond_max)
        guarded.append((self.ref(bn), ok))
        return ok or not guard_on

    def recheck():
        for r, was in guarded:
            ok, vals = half_affine_ok(self.get(r), eps, cond_max)
            if ok != was and guard_on:
                from .engine import PtxError
                raise PtxError("BigGAN fp16 plan: a BatchNorm's folded tables %s the range the packed-fp16 affine is safe in "
                               "(max |scale| %.3g, max |shift| %.3g, max |mean| / sigma %.3g) since this plan was compiled: call "
                               "model.refresh() to recompile it" % ("left" if was else "entered", vals[0], vals[1], vals[2]))
    if torch.device(self.dev).type != "meta":
        self.refreshers.append(recheck)

    def pro_ok(nb, H, W):
        """The NEXT block's conv1 can apply its own cBN1 + ReLU to its input fragments (ptx_conv1x1_pro_f16_fwd), so the conv
        that produces its input stores the raw sum only.  PTX_CONV1_PRO=0: the two-output flow (A/B runs)."""
        if nb is None or nb.kind != "gblock" or os.environ.get("PTX_CONV1_PRO", "1") == "0":
            return False
        K, Co = nb.conv1.in_channels, nb.conv1.out_channels
        # (16 x 16 maps stay on the two-output flow: 64 workgroups of 16 serial chunks measured 0.031 vs 0.020 ms)
        if not (K % 128 == 0 and 128 <= K <= 2048 and Co in (64, 128, 256, 512) and (H * W) % 256 == 0 and H * W >= 1024):
            return False
        # the PRODUCER of this map stores the raw sum only on a "yes", so the answer must be the consumer's own: ask the
        # library with the descriptor Plan.conv will build for that conv1 (extents, strides, 2 GiB limits included)
        from ._lib import ConvDesc, PTX_EPI_AFFINE, PTX_EPI_OUT_F16, PTX_EPI_RELU, PTX_F16_OPERANDS
        pk_ = self.pack(nb.conv1, None, f16=True)
        d = ConvDesc()
        ld_in, ld_out = (K + 7) // 8 * 8, (Co + 7) // 8 * 8
        d.N, d.Ti, d.Hi, d.Wi, d.Ci, d.ldx = h.N, 1, H, W, K // 2, ld_in // 2
        d.To, d.Ho, d.Wo, d.Co, d.ldy = 1, H, W, Co, ld_out
        d.kT = d.kH = d.kW = d.sT = d.sH = d.sW = 1
        d.Kc, d.Co_pad, d.groups = pk_.Kc // 2, pk_.Co_pad, 1
        d.flags = PTX_F16_OPERANDS | PTX_EPI_OUT_F16 | PTX_EPI_AFFINE | PTX_EPI_RELU
        if not self.lib.ptx_conv1x1_pro_f16_supported(C.byref(d)):
            return False
        # ... and the tables that kernel would apply as packed fp16 FMAs must be in range and well conditioned
        return half_ok(nb.bn1)

    xa, xr = cbn(h, flat[0][2].bn1), h          # activated input of the first block (halfs), its skip operand (fp32)
    for k, (si, bi, blk) in enumerate(flat):
        name = "blocks.%d.%d" % (si, bi)
        nxt = flat[k + 1][2] if k + 1 < len(flat) else None
        if blk.kind != "gblock" and getattr(xr, "f16", False) and nxt is not None and nxt.kind == "gblock":
            # self-attention inside the half chain (round 4): projections on the fp16 tiles, the output conv + residual + the
            # next block's cBN1 + ReLU as ONE launch that writes both things the next block reads (or the raw sum alone when
            # the next conv1 activates its own input)
            xa, xr = biggan_attention_f16(self, xr, blk, name, None if pro_ok(nxt, xr.H, xr.W) else tab(nxt.bn1))
            continue
        if blk.kind != "gblock":                 # self-attention on the fp32 raw map, then re-enter the fp16 chain
            h32 = biggan_attention(self, xr, blk, name)
            xa, xr = (cbn(h32, nxt.bn1), h32) if nxt is not None else (None, h32)
            continue
        up = bool(blk.upsample)
        pk = lambda c: self.pack(c, None, f16=True)            # noqa: E731
        if xa is None:                           # raw input: cBN1 + ReLU on this conv's input fragments
            t = self.conv(xr, pk(blk.conv1), one, zero, relu=True, affine=tab(blk.bn2), out_f16=True, pro_affine=tab(blk.bn1),
                          label=name + ".conv1")
        else:
            t = self.conv(xa, pk(blk.conv1), one, zero, relu=True, affine=tab(blk.bn2), out_f16=True, label=name + ".conv1")
        t = self.conv(t, pk(blk.conv2), one, (0, 1, 1), relu=True, affine=tab(blk.bn3), out_f16=True, up2=up,
                      label=name + ".conv2")
        t = self.conv(t, pk(blk.conv3), one, (0, 1, 1), relu=True, affine=tab(blk.bn4), out_f16=True, label=name + ".conv3")
        skip = dict(res=xr) if (not up and blk.in_channels == blk.out_channels) else \
            dict(res=xr, res_kind="up", res_stride=(0, int(up), int(up)))
        if nxt is None and _rgb_conv_ok(self, blk.conv4.out_channels, t, model.output_layer[2], obn.channels) and half_ok(obn):
            # last block, image conv on its own kernel (round 4): the output layer's BN + ReLU runs on that kernel's A
            # fragments, so this conv stores the RAW sum only -- no activated copy of the last feature map exists
            xr = self.conv(t, pk(blk.conv4), one, zero, out_f16=True, label=name + ".conv4", **skip)
            xa = None
            self.feat = _rgb_conv(self, xr, model.output_layer[2], _ptr(oscale), _ptr(oshift), obn.channels)
            self.pooled = None
            return
        if nxt is None:                          # last block: the output layer's BN + ReLU in the epilogue
            xa = self.conv(t, pk(blk.conv4), one, zero, relu=True, affine=(_ptr(oscale), _ptr(oshift), obn.channels),
                           out_f16=True, label=name + ".conv4", **skip)
            xr = None
        elif pro_ok(nxt, t.H, t.W) and getattr(xr, "f16", False):
            # (only where the skip operand already is halfs: the first block's skip is the fp32 linear output and keeps the
            #  two-output fused stage, so the chain of halfs starts there)
            # the next block activates its own input: ONE output, the raw sum (half of what this conv used to write)
            xr = self.conv(t, pk(blk.conv4), one, zero, out_f16=True, label=name + ".conv4", **skip)
            xa = None
        elif nxt.kind == "gblock":               # next block's cBN1 + ReLU, plus the raw sum for its skip
            xa, xr = self.conv(t, pk(blk.conv4), one, zero, relu=True, affine=tab(nxt.bn1), out_f16=True, raw=True,
                               label=name + ".conv4", **skip)
        elif _half_attention_ok(flat, k + 1):    # attention next, inside the half chain: the raw sum as halfs
            xr = self.conv(t, pk(blk.conv4), one, zero, out_f16=True, label=name + ".conv4", **skip)
            xa = None
        else:                                    # attention next: it wants the raw fp32 map
            xr = self.conv(t, pk(blk.conv4), one, zero, label=name + ".conv4", **skip)
            xa = None
    if xa is None:                               # the network ended on an attention block (not a published layout)
        xa = affine(xr, _ptr(oscale), _ptr(oshift), obn.channels, 1)
    self.feat = self.conv(xa, self.pack(model.output_layer[2], None, f16=True), one, (0, 1, 1), tanh=True,
                          label="output_layer.2")
    self.pooled = None

def _rgb_conv_ok(self, channels, x=None, conv=None, ld_aff=0):
    """The generator's image conv has its own kernel for C in {32, 64, 128} (ptx_rgb_conv3x3_f16_fwd); PTX_RGB_CONV=0 keeps
    the implicit-GEMM tile (A/B runs).  `x` (the map the last block will produce), `conv`, `ld_aff`: the decision is then the
    library's own answer for the descriptor _rgb_conv will pass -- the last block stores a raw-only map on a "yes"."""
    import os
    if os.environ.get("PTX_RGB_CONV", "1") == "0" or channels not in (32, 64, 128):
        return False
    if x is None:
        return True
    from ._lib import RgbConvDesc, PTX_EPI_TANH
    if conv.out_channels != 3 or tuple(conv.kernel_size) != (3, 3) or tuple(conv.padding) != (1, 1):
        return False
    d = RgbConvDesc(x.N, x.H, x.W, channels, (channels + 7) // 8 * 8, 4, ld_aff, PTX_EPI_TANH)
    return bool(self.lib.ptx_rgb_conv3x3_f16_supported(C.byref(d)))


def _rgb_conv(self, x, conv, scale_ptr, shift_ptr, ld_aff):
    """BN -> ReLU -> conv3x3(C -> 3) -> tanh as ONE launch on the raw half feature map x (gen_stage_f16.hip)."""
    from ._lib import RgbConvDesc, PTX_EPI_TANH
    from .engine import PtxError, _tag
    lib = self.lib
    d = RgbConvDesc(x.N, x.H, x.W, x.C, x.ld, 4, ld_aff, PTX_EPI_TANH)
    if not x.f16 or conv.out_channels != 3 or tuple(conv.kernel_size) != (3, 3) or tuple(conv.padding) != (1, 1) \
            or not lib.ptx_rgb_conv3x3_f16_supported(C.byref(d)):
        raise PtxError("output layer: not the 3x3, 3-channel image conv the fused kernel covers")
    y = self.act(x.N, 1, x.H, x.W, 3)
    wp = torch.empty(int(lib.ptx_rgb_conv_weight_elems(x.C)), device=self.dev, dtype=torch.float16)
    bias = torch.zeros(4, device=self.dev, dtype=torch.float32)
    self.keepalive += [d, wp, bias]
    ref = self.ref(conv)

    def refresh():
        cv = self.get(ref)
        w = cv.weight.detach().contiguous()
        check(lib.ptx_pack_rgb_conv_weight(_ptr(w), x.C, C.c_void_p(wp.data_ptr()), _stream()), "ptx_pack_rgb_conv_weight")
        bias.zero_()
        if cv.bias is not None:
            bias[:3].copy_(cv.bias.detach())
    if torch.device(self.dev).type != "meta":
        self.refreshers.append(refresh)
    xp, yp, wpp, bp = C.c_void_p(x.t.data_ptr()), _ptr(y.t), C.c_void_p(wp.data_ptr()), _ptr(bias)

    def step(st):
        check(lib.ptx_rgb_conv3x3_f16_fwd(C.byref(d), xp, scale_ptr, shift_ptr, wpp, bp, yp, st), "ptx_rgb_conv3x3_f16_fwd")
    self.steps.append(_tag(step, "rgb_conv3x3", 2 * x.N * x.H * x.W * x.C + 16 * x.N * x.H * x.W))
    return y


def _half_attention_ok(flat, k):
    """The attention block flat[k] can run inside the half chain: a GBlock follows it (whose cBN1 its output conv folds) and
    its widths fit the half kernels (theta / phi d = ch / 8 <= 64: the fp16-operand attention; ch / 2 in {64, 128, 256} and
    ch a multiple of 128: ptx_conv1x1_skip_f16_fwd).  PTX_ATTN_F16=0 keeps the fp32 block (A/B runs)."""
    import os
    if os.environ.get("PTX_ATTN_F16", "1") == "0" or k + 1 >= len(flat) or flat[k + 1][2].kind != "gblock":
        return False
    ch = flat[k][2].ch
    return ch // 8 <= 64 and ch // 2 in (64, 128, 256) and ch % 128 == 0


def biggan_attention_f16(self, x, att, name, next_affine):
    """layers.Attention on a HALF raw map x, returning (relu(cBN1_next(out)) halfs, out halfs): theta / phi / g on the fp16-
    operand tiles (fp32 out: the attention kernel rounds its operands to halfs in registers anyway), 2 x 2 max pools, the
    fused attention kernel writing halfs (PTX_NL_OUT_F16), and `o` * gamma + x + the next block's cBN1 + ReLU with both
    outputs as one ptx_conv1x1_skip_f16_fwd launch -- the fp32 round trip of the block, its x3 convs and the separate
    affine pass behind it are gone."""
    c8, c2 = att.ch // 8, att.ch // 2
    one, zero = (1, 1, 1), (0, 0, 0)
    tpg = self.conv(x, self.pack([att.theta, att.phi, att.g], None, f16=True), one, zero, label=name + ".theta_phi_g")
    phi = self.maxpool(tpg.slice(c8, c8), (1, 2, 2), (1, 2, 2), (0, 0, 0))
    g = self.maxpool(tpg.slice(2 * c8, c2), (1, 2, 2), (1, 2, 2), (0, 0, 0))
    yatt = self.act(x.N, 1, x.H, x.W, c2, f16=True)
    if not self.attention(tpg.slice(0, c8), phi, g, yatt, f16=True):
        raise PtxError("%s: the fused fp16 attention refused a shape _half_attention_ok admitted" % name)
    pko = self.pack(att.o, None, scale=(att, "gamma"), f16=True)
    if next_affine is None:                      # the next conv1 activates its own input: the raw sum only
        return None, self.conv(yatt, pko, one, zero, out_f16=True, res=x, label=name + ".o")
    return self.conv(yatt, pko, one, zero, relu=True, affine=next_affine, out_f16=True, raw=True, res=x, label=name + ".o")


def biggan_attention(self, x, att, name):
    """layers.Attention: theta^T phi over 2x2-max-pooled keys, softmax, values g, output conv * gamma + x."""
    lib = self.lib
    N, HW = x.N, x.H * x.W
    c8, c2 = att.ch // 8, att.ch // 2
    one, zero = (1, 1, 1), (0, 0, 0)
    # the attention block works on the fp32 raw map (softmax wants fp32 logits); in the fp16 generator plan its two
    # pointwise convs run as split operands on the fp16 matrix cores (fp32-accurate, ~2.5x the fp32-MFMA rate)
    x3 = True if getattr(self, "half_plan", False) else None
    tpg = self.conv(x, self.pack([att.theta, att.phi, att.g], None, x3=x3), one, zero, label=name + ".theta_phi_g")
    phi = self.maxpool(tpg.slice(c8, c8), (1, 2, 2), (1, 2, 2), (0, 0, 0))
    g = self.maxpool(tpg.slice(2 * c8, c2), (1, 2, 2), (1, 2, 2), (0, 0, 0))
    S4 = HW // 4
    yatt = self.act(N, 1, x.H, x.W, c2)
    if self.attention(tpg.slice(0, c8), phi, g, yatt, f16=bool(getattr(self, "half_plan", False))):
        return self.conv(yatt, self.pack(att.o, None, scale=(att, "gamma"), x3=x3), one, zero, res=x, label=name + ".o")
    ldf = _r4(S4)
    f = torch.empty((N, HW, ldf), device=self.dev, dtype=torch.float32)
    gT = torch.empty((N, c2, ldf), device=self.dev, dtype=torch.float32)
    self.keepalive += [f, gT]
    th, ph, gp, fp, gtp, yp = _ptr(tpg.t), _ptr(phi.t), _ptr(g.t), _ptr(f), _ptr(gT), _ptr(yatt.t)
    ld3, ldp, ldg, yld = tpg.ld, phi.ld, g.ld, yatt.ld

    def step(st):
        check(lib.ptx_bgemm_nt(th, ph, fp, N, HW, S4, c8, ld3, ldp, ldf, HW * ld3, S4 * ldp, HW * ldf, st), "attn f")
        check(lib.ptx_softmax_rows(fp, N * HW, S4, ldf, 0, st), "attn softmax")
        check(lib.ptx_transpose_last2(gp, gtp, N, S4, c2, ldg, ldf, st), "attn g^T")
        check(lib.ptx_bgemm_nt(fp, gtp, yp, N, HW, c2, S4, ldf, ldf, yld, HW * ldf, c2 * ldf, HW * yld, st), "attn y")
    self.steps.append(step)
    return self.conv(yatt, self.pack(att.o, None, scale=(att, "gamma")), one, zero, res=x, label=name + ".o")
