"""Autograd bridge: the reference's nn.Module forward/backward (model/DCGAN.py:29-35,61-67) expressed as
torch.autograd.Functions over the C-ABI kernels.  Tensors between stages are NHWC in the library's element type
(bf16 or fp32, `prec`); NCHW fp32 exists only at the module boundary.

The layer Functions are first-order (`once_differentiable`).  The one second-order quantity the reference forms - the CGAN
gradient penalty, back-propagated through `autograd.grad(..., create_graph=True)` (train/cgan_trainer.py:114-131,200-203) - is
provided as ONE differentiable Function, `gradient_penalty`, whose backward is the closed-form double backward of the step engine
(PHASE_GP_ONLY).  No CPU path: every Function requires device tensors.
"""
import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from ._lib import PREC_BF16, PREC_F32, JckError, cur_stream, lib

_PREC = {"bf16": PREC_BF16, "f32": PREC_F32}
_DT = {PREC_BF16: torch.bfloat16, PREC_F32: torch.float32}
BN_MOMENTUM, BN_EPS = 0.1, 1e-5


def _need_cuda(t, what):
    if not t.is_cuda:
        raise JckError(f"{what}: got a {t.device} tensor - the MI355X path has no CPU fallback; move the module and its "
                       f"inputs to the GPU")


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


def _pack(kind, w, prec, *dims):
    """fp32 parameter -> GEMM operand of the precision's element type (include/jckgan.h, weight packing)."""
    a, b = w.shape[0], w.shape[1]
    if kind == "down":
        n = lib.jck_pad_rows(a) * 16 * lib.jck_pad_chan(b)
    elif kind == "up":
        n = 4 * lib.jck_pad_rows(b) * 4 * a
    else:  # g1
        n = 16 * b * dims[0]
    wp = torch.empty(n, dtype=_DT[prec], device=w.device)
    wf = _f32(w)
    if kind == "down":
        lib.jck_pack_down(prec, wf, a, b, wp, cur_stream())
    elif kind == "up":
        lib.jck_pack_up(prec, wf, a, b, wp, cur_stream())
    else:
        lib.jck_pack_g1(prec, wf, a, b, dims[0], wp, cur_stream())
    return wp


def _stats_buf(pixels, c, nyrep, dev):
    return torch.empty(lib.jck_stats_floats(pixels, c, nyrep), dtype=torch.float32, device=dev)


class _ToNHWC(Function):
    """NCHW fp32 image -> NHWC4 T."""

    @staticmethod
    def forward(ctx, x, prec):
        ctx.prec = prec
        n, c, h, w = x.shape
        out = torch.empty(n, h, w, 4, dtype=_DT[prec], device=x.device)
        lib.jck_img_prep(prec, _f32(x), None, 1.0, 0.0, out, n, h * w, cur_stream())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n, h, w, _ = g.shape
        out = torch.empty(n, 3, h, w, dtype=torch.float32, device=g.device)
        lib.jck_nhwc4_to_nchw(ctx.prec, g.contiguous(), out, n, h * w, cur_stream())
        return out, None


class _ToNCHW(Function):
    """NHWC4 T -> NCHW fp32 image."""

    @staticmethod
    def forward(ctx, x, prec):
        ctx.prec = prec
        n, h, w, _ = x.shape
        out = torch.empty(n, 3, h, w, dtype=torch.float32, device=x.device)
        lib.jck_nhwc4_to_nchw(prec, x.contiguous(), out, n, h * w, cur_stream())
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n, c, h, w = g.shape
        out = torch.empty(n, h, w, 4, dtype=_DT[ctx.prec], device=g.device)
        lib.jck_img_prep(ctx.prec, _f32(g), None, 1.0, 0.0, out, n, h * w, cur_stream())
        return out, None


class _ConvBnAct(Function):
    """One stage: {Conv2d k4s2p1 | ConvTranspose2d k4s2p1 | ConvTranspose2d k4s1p0 on 1x1} -> BatchNorm2d(train) ->
    ReLU/LeakyReLU.  BN statistics come out of the GEMM epilogue; running stats are updated in place."""

    @staticmethod
    def forward(ctx, x, w, gamma, beta, rm, rv, nbt, kind, slope, prec):
        dev = x.device
        st = cur_stream()
        x = x.contiguous()
        if kind == "down":          # x [N,Hb,Wb,CbPad] -> [N,Hb/2,Wb/2,Cs]
            n, hb, wb, _ = x.shape
            cs, cb = w.shape[0], w.shape[1]
            oshape, rows, c = (n, hb // 2, wb // 2, cs), n * (hb // 2) * (wb // 2), cs
        elif kind == "up":          # x [N,Hs,Ws,Cs] -> [N,2Hs,2Ws,Cb]
            n, hs, ws_, cs = x.shape
            cb = w.shape[1]
            oshape, rows, c = (n, 2 * hs, 2 * ws_, cb), n * 4 * hs * ws_, cb
        else:                       # g1: x [B,CiPad] -> [B,4,4,Co]
            n, cip = x.shape
            c = w.shape[1]
            oshape, rows = (n, 4, 4, c), n * 16
        y = torch.empty(oshape, dtype=_DT[prec], device=dev)
        stats = _stats_buf(rows, c, 16 if kind == "g1" else 1, dev)
        slots = ctypes.c_int(0)
        if kind == "down":
            lib.jck_conv_down(prec, x, _pack("down", w, prec), y, stats, ctypes.byref(slots), n, hb, wb, cb, cs, st)
        elif kind == "up":
            lib.jck_conv_up(prec, x, _pack("up", w, prec), y, stats, ctypes.byref(slots), 0, n, hs, ws_, cs, cb, st)
        else:
            lib.jck_g1_fwd(prec, x, _pack("g1", w, prec, cip), y, stats, ctypes.byref(slots), n, cip, c, st)
        aux = torch.empty(4 * c, dtype=torch.float32, device=dev)
        lib.jck_bn_finalize(stats, slots.value, float(rows), _f32(gamma), _f32(beta), rm, rv, nbt, BN_MOMENTUM, BN_EPS, aux, c,
                            st)
        a = torch.empty_like(y)
        lib.jck_bn_act_fwd(prec, y, aux, slope, a, rows, c, st)
        ctx.save_for_backward(x, w, y, aux)
        ctx.meta = (kind, slope, prec, rows, c)
        return a

    @staticmethod
    @once_differentiable
    def backward(ctx, ga):
        x, w, y, aux = ctx.saved_tensors
        kind, slope, prec, rows, c = ctx.meta
        dev, st = x.device, cur_stream()
        f32 = dict(dtype=torch.float32, device=dev)
        sums = torch.empty(lib.jck_bn_bwd_ws_floats(c), **f32)
        dgam, dbet = torch.zeros(c, **f32), torch.zeros(c, **f32)
        gy = torch.empty_like(y)
        lib.jck_bn_act_bwd(prec, ga.contiguous(), y, aux, slope, sums, gy, dgam, dbet, rows, c, st)
        gw = gx = None
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if kind == "down":
            n, hb, wb, _ = x.shape
            cs, cb = w.shape[0], w.shape[1]
            if need_w:
                nb = lib.jck_conv_wgrad_ws_bytes(n, hb, wb, cb, cs)
                ws = torch.empty(nb // 4, **f32)
                gw = torch.empty(w.shape, **f32)
                lib.jck_conv_wgrad(prec, gy, x, ws, nb, gw, 0, n, hb, wb, cb, cs, st)
            if need_x:
                gx = torch.empty_like(x)
                lib.jck_conv_up(prec, gy, _pack("up", w, prec), gx, None, None, 0, n, hb // 2, wb // 2, cs, cb, st)
        elif kind == "up":
            n, hs, ws_, cs = x.shape
            cb = w.shape[1]
            if need_w:
                nb = lib.jck_conv_wgrad_ws_bytes(n, 2 * hs, 2 * ws_, cb, cs)
                ws = torch.empty(nb // 4, **f32)
                gw = torch.empty(w.shape, **f32)
                lib.jck_conv_wgrad(prec, x, gy, ws, nb, gw, 0, n, 2 * hs, 2 * ws_, cb, cs, st)
            if need_x:
                gx = torch.empty_like(x)
                lib.jck_conv_down(prec, gy, _pack("down", w, prec), gx, None, None, n, 2 * hs, 2 * ws_, cb, cs, st)
        else:
            n, cip = x.shape
            ci = w.shape[0]
            if need_w:
                nb = lib.jck_g1_wgrad_ws_bytes(n, cip, c)
                ws = torch.empty(nb // 4, **f32)
                gw = torch.empty(w.shape, **f32)
                lib.jck_g1_wgrad(prec, x, gy, ws, nb, gw, 0, n, ci, cip, c, st)
            if need_x:
                raise JckError("gradient w.r.t. the latent input of G.conv1 is not provided")
        return gx, gw, dgam, dbet, None, None, None, None, None, None


class _UpTanh(Function):
    """G.conv5 + tanh (model/DCGAN.py:58-59,66): [N,32,32,64] -> NHWC4 image in (-1,1)."""

    @staticmethod
    def forward(ctx, x, w, prec):
        n, hs, ws_, cs = x.shape
        cb = w.shape[1]
        x = x.contiguous()
        y = torch.empty(n, 2 * hs, 2 * ws_, lib.jck_pad_chan(cb), dtype=_DT[prec], device=x.device)
        lib.jck_conv_up(prec, x, _pack("up", w, prec), y, None, None, 1, n, hs, ws_, cs, cb, cur_stream())
        ctx.save_for_backward(x, w, y)
        ctx.prec = prec
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        prec, st = ctx.prec, cur_stream()
        n, hs, ws_, cs = x.shape
        cb = w.shape[1]
        graw = torch.empty_like(y)
        lib.jck_tanh_bwd(prec, g.contiguous(), y, 1.0, graw, y.numel(), st)
        gw = gx = None
        if ctx.needs_input_grad[1]:
            nb = lib.jck_conv_wgrad_ws_bytes(n, 2 * hs, 2 * ws_, cb, cs)
            ws = torch.empty(nb // 4, dtype=torch.float32, device=x.device)
            gw = torch.empty(w.shape, dtype=torch.float32, device=x.device)
            lib.jck_conv_wgrad(prec, x, graw, ws, nb, gw, 0, n, 2 * hs, 2 * ws_, cb, cs, st)
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            lib.jck_conv_down(prec, graw, _pack("down", w, prec), gx, None, None, n, 2 * hs, 2 * ws_, cb, cs, st)
        return gx, gw, None


class _HeadSigmoid(Function):
    """D.conv5 + sigmoid (model/DCGAN.py:26-27,34): [N,4,4,512] -> [N,1,1,1] probability (fp32)."""

    @staticmethod
    def forward(ctx, a4, w, prec):
        n = a4.shape[0]
        c = w.shape[1]
        dev = a4.device
        a4 = a4.contiguous()
        wp = torch.empty(16 * c, dtype=torch.float32, device=dev)
        lib.jck_pack_head(_f32(w), c, wp, cur_stream())
        prob = torch.empty(n, dtype=torch.float32, device=dev)
        dsig = torch.empty(n, dtype=torch.float32, device=dev)          # p(1-p)
        lib.jck_head_fwd(prec, a4, wp, None, n, 16 * c, 0.0, 1, prob, dsig, None, -1, -1, 0, cur_stream())
        ctx.save_for_backward(a4, wp, dsig)
        ctx.meta = (prec, c)
        return prob.view(n, 1, 1, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, gp):
        a4, wp, dsig = ctx.saved_tensors
        prec, c = ctx.meta
        n = a4.shape[0]
        ds = (gp.reshape(n).to(torch.float32) * dsig).contiguous()
        ga = torch.empty_like(a4) if ctx.needs_input_grad[0] else None
        gw = None
        dwp = torch.empty(16 * c, dtype=torch.float32, device=a4.device) if ctx.needs_input_grad[1] else None
        hws = torch.empty(lib.jck_head_bwd_ws_floats(16 * c), dtype=torch.float32, device=a4.device) if dwp is not None else None
        lib.jck_head_bwd(prec, ds, wp, a4, n, 16 * c, ga, dwp, 0, hws, cur_stream())
        if dwp is not None:
            gw = torch.empty(1, c, 4, 4, dtype=torch.float32, device=a4.device)
            lib.jck_head_unpack_grad(dwp, c, gw, 0, cur_stream())
        return ga, gw, None


def _stage_eval(conv, norm, x, kind, slope, prec):
    """module.eval(): normalise with the RUNNING statistics and leave the buffers alone (nn.BatchNorm2d semantics).  The
    reference never puts G / D in eval mode; this exists so that a checkpoint loaded for inference with
    .eval() behaves like the torch modules it is interchangeable with.  Inference only (no autograd through it)."""
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad):
        raise JckError("eval-mode BatchNorm is provided for inference only: wrap the call in torch.no_grad() (training runs "
                       "with batch statistics, as the reference always does)")
    dev, st, w = x.device, cur_stream(), conv.weight
    x = x.contiguous()
    if kind == "down":
        n, hb, wb, _ = x.shape
        cs, cb = w.shape[0], w.shape[1]
        y = torch.empty(n, hb // 2, wb // 2, cs, dtype=_DT[prec], device=dev)
        lib.jck_conv_down(prec, x, _pack("down", w, prec), y, None, None, n, hb, wb, cb, cs, st)
        rows, c = n * (hb // 2) * (wb // 2), cs
    elif kind == "up":
        n, hs, ws_, cs = x.shape
        cb = w.shape[1]
        y = torch.empty(n, 2 * hs, 2 * ws_, cb, dtype=_DT[prec], device=dev)
        lib.jck_conv_up(prec, x, _pack("up", w, prec), y, None, None, 0, n, hs, ws_, cs, cb, st)
        rows, c = n * 4 * hs * ws_, cb
    else:
        n, cip = x.shape
        c = w.shape[1]
        y = torch.empty(n, 4, 4, c, dtype=_DT[prec], device=dev)
        lib.jck_g1_fwd(prec, x, _pack("g1", w, prec, cip), y, None, None, n, cip, c, st)
        rows = n * 16
    invstd = torch.rsqrt(norm.running_var.float() + norm.eps)
    sc = norm.weight.detach().float() * invstd
    aux = torch.cat([sc, norm.bias.detach().float() - norm.running_mean.float() * sc, norm.running_mean.float(), invstd]).contiguous()
    a = torch.empty_like(y)
    lib.jck_bn_act_fwd(prec, y, aux, slope, a, rows, c, st)
    return a


def _stage(mod, i, x, kind, slope, prec):
    conv, norm = getattr(mod, f"conv{i}"), getattr(mod, f"norm{i}")
    if not mod.training:
        return _stage_eval(conv, norm, x, kind, slope, prec)
    return _ConvBnAct.apply(x, conv.weight, norm.weight, norm.bias, norm.running_mean, norm.running_var,
                            norm.num_batches_tracked, kind, slope, prec)


def _n_stages(mod):
    """BatchNorm stages of a DCGAN module: 4 (the reference's 64x64 nets) or 5 (the 128x128 plan)."""
    n = 0
    while hasattr(mod, f"norm{n + 1}"):
        n += 1
    return n


def dcgan_discriminator(mod, x, prec="bf16"):
    """model/DCGAN.py:29-35 on the HIP path.  x: NCHW fp32 [B,3,S,S] on the GPU -> [B,1,1,1] fp32 (S = 64, or 128 for a
    module built with image_size=128)."""
    _need_cuda(x, "Discriminator.forward")
    n = _n_stages(mod)
    size = 4 << n
    if x.dim() != 4 or x.shape[1:] != (3, size, size):
        raise JckError(f"Discriminator expects [B,3,{size},{size}], got {tuple(x.shape)}")
    p = _PREC[prec]
    h = _ToNHWC.apply(x, p)
    for i in range(1, n + 1):
        h = _stage(mod, i, h, "down", 0.2, p)
    return _HeadSigmoid.apply(h, getattr(mod, f"conv{n + 1}").weight, p)


def dcgan_generator(mod, z, prec="bf16"):
    """model/DCGAN.py:61-67 on the HIP path.  z: [B,100,1,1] fp32 on the GPU -> NCHW fp32 [B,3,64,64]."""
    _need_cuda(z, "Generator.forward")
    if z.dim() != 4 or z.shape[1:] != (mod.conv1.weight.shape[0], 1, 1):
        raise JckError(f"Generator expects [B,{mod.conv1.weight.shape[0]},1,1], got {tuple(z.shape)}")
    p = _PREC[prec]
    b, ci = z.shape[0], z.shape[1]
    cip = 128 if ci <= 128 else 256
    zp = torch.zeros(b, cip, dtype=_DT[p], device=z.device)
    zp[:, :ci] = z.reshape(b, ci).to(_DT[p])
    n = _n_stages(mod)
    h = _stage(mod, 1, zp, "g1", 0.0, p)
    for i in range(2, n + 1):
        h = _stage(mod, i, h, "up", 0.0, p)
    img = _UpTanh.apply(h, getattr(mod, f"conv{n + 1}").weight, p)
    return _ToNCHW.apply(img, p)


# --------------------------------------------------------------------------------------------------------------------
# CGAN head (model/CGAN.py:109-123): label MLP, concat, Linear(8392,256), Dropout(0.25), Linear(256,1), sigmoid
# --------------------------------------------------------------------------------------------------------------------
L1_K, L1_KPAD, L1_OUT, FEAT, EMB, L1_KSPLIT = 8392, 8448, 256, 8192, 200, 12


class _CganHead(Function):
    """a4 [B,4,4,512] (NHWC), one-hot labels -> probability [B,1].  First-order backward to a4 and to the five parameter
    tensors.  `mask` is the dropout keep mask [B,256] (float 0/1)."""

    @staticmethod
    def forward(ctx, a4, labels, mask, we, be, w1, b1, w2, b2, prec):
        dev, st = a4.device, cur_stream()
        B = a4.shape[0]
        dt = _DT[prec]
        a4 = a4.contiguous()
        cbuf = torch.zeros(B, L1_KPAD, dtype=dt, device=dev)
        lib.jck_concat_rows(prec, a4, FEAT, cbuf, L1_KPAD, B, st)
        pre_e = torch.empty(B * EMB, dtype=torch.float32, device=dev)
        lab = labels.to(torch.int64).contiguous()
        lib.jck_label_embed_fwd(prec, lab, _f32(we), _f32(be), 0.2, B, 100, EMB, cbuf, L1_KPAD, FEAT, pre_e, st)
        w1p = torch.empty(L1_OUT * L1_KPAD, dtype=dt, device=dev)
        lib.jck_pack_linear(prec, _f32(w1), L1_OUT, L1_K, L1_OUT, L1_KPAD, 0, 512, 16, w1p, st)
        slab = torch.empty(L1_KSPLIT * B * L1_OUT, dtype=torch.float32, device=dev)
        lib.jck_linear_fwd(prec, cbuf, w1p, None, slab, B, L1_KPAD, L1_OUT, L1_OUT, L1_KSPLIT, st)
        hd = torch.empty(B, L1_OUT, dtype=dt, device=dev)
        mask = mask.to(torch.float32).contiguous()
        lib.jck_linear_finish(prec, slab, L1_KSPLIT, _f32(b1), mask, 1.0 / 0.75, None, hd, B, L1_OUT, st)
        prob = torch.empty(B, dtype=torch.float32, device=dev)
        dsig = torch.empty(B, dtype=torch.float32, device=dev)
        w2f, b2f = _f32(w2).view(-1), _f32(b2)
        lib.jck_head_fwd(prec, hd, w2f, b2f, B, L1_OUT, 0.0, 1, prob, dsig, None, -1, -1, 0, st)
        ctx.save_for_backward(a4, lab, mask, cbuf, pre_e, hd, dsig, w1, w2f)
        ctx.prec = prec
        return prob.view(B, 1)

    @staticmethod
    @once_differentiable
    def backward(ctx, gp):
        a4, lab, mask, cbuf, pre_e, hd, dsig, w1, w2f = ctx.saved_tensors
        prec, st = ctx.prec, cur_stream()
        dev, dt = a4.device, _DT[prec]
        B = a4.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        ds = (gp.reshape(B).to(torch.float32) * dsig).contiguous()
        ghd = torch.empty(B, L1_OUT, dtype=dt, device=dev)
        gw2 = torch.zeros(L1_OUT, **f32)
        hws = torch.empty(lib.jck_head_bwd_ws_floats(L1_OUT), **f32)
        lib.jck_head_bwd(prec, ds, w2f, hd, B, L1_OUT, ghd, gw2, 1, hws, st)
        gb2 = torch.zeros(1, **f32)
        lib.jck_sum_vec(ds, B, gb2, st)
        gh = torch.empty_like(ghd)
        lib.jck_dropout(prec, ghd, mask, 1.0 / 0.75, gh, B * L1_OUT, st)
        nb = lib.jck_linear_wgrad_ws_bytes(B, L1_KPAD, L1_OUT)
        ws = torch.empty(nb // 4, **f32)
        gw1p = torch.empty(L1_OUT * L1_KPAD, **f32)
        lib.jck_linear_wgrad(prec, gh, L1_OUT, cbuf, L1_KPAD, ws, nb, gw1p, 0, B, L1_OUT, st)
        gw1 = torch.empty(L1_OUT, L1_K, **f32)
        lib.jck_unperm_linear_grad(gw1p, L1_OUT, L1_K, L1_KPAD, 512, 16, gw1, 0, st)
        gb1 = torch.zeros(L1_OUT, **f32)
        lib.jck_colsum(prec, gh, B, L1_OUT, L1_OUT, gb1, st)
        w1t = torch.empty(L1_KPAD * L1_OUT, dtype=dt, device=dev)
        lib.jck_pack_linear(prec, _f32(w1), L1_OUT, L1_K, L1_KPAD, L1_OUT, 1, 512, 16, w1t, st)
        gc = torch.empty(B, L1_KPAD, dtype=dt, device=dev)
        lib.jck_linear_fwd(prec, gh, w1t, None, gc, B, L1_OUT, L1_KPAD, L1_KPAD, 1, st)
        ga4 = torch.empty_like(a4)
        lib.jck_split_rows(prec, gc, L1_KPAD, FEAT, ga4, B, st)
        gwe, gbe = torch.zeros(EMB, 100, **f32), torch.zeros(EMB, **f32)
        lib.jck_label_embed_bwd(prec, gc, L1_KPAD, FEAT, pre_e, lab, 0.2, B, 100, EMB, gwe, gbe, st)
        return ga4, None, None, gwe, gbe, gw1, gb1, gw2.view(1, L1_OUT), gb2, None


def cgan_discriminator(mod, x, labels, prec="bf16", mask=None):
    """model/CGAN.py:109-123 on the HIP path.  x NCHW fp32 [B,3,64,64] + one-hot int64 [B,100] -> [B,1] fp32."""
    _need_cuda(x, "Discriminator.forward")
    if x.dim() != 4 or x.shape[1:] != (3, 64, 64) or labels.shape != (x.shape[0], 100):
        raise JckError(f"Discriminator expects [B,3,64,64] and one-hot [B,100], got {tuple(x.shape)}, {tuple(labels.shape)}")
    p = _PREC[prec]
    h = _ToNHWC.apply(x, p)
    for i in (1, 2, 3, 4):
        h = _stage(mod, i, h, "down", 0.2, p)
    if mask is None:        # nn.Dropout(0.25) in training mode (the reference never switches D to eval)
        mask = (torch.rand(x.shape[0], L1_OUT, device=x.device) >= 0.25).float() if mod.training else \
            torch.full((x.shape[0], L1_OUT), 0.75, device=x.device)
    return _CganHead.apply(h, labels.to(x.device), mask, mod.label_embedding.weight, mod.label_embedding.bias,
                           mod.linear1.weight, mod.linear1.bias, mod.linear2.weight, mod.linear2.bias, p)


# ---------------------------------------------------------------------------------------------------------------------
# the gradient penalty as a differentiable quantity (train/dcgan_trainer.py:110-127, train/cgan_trainer.py:114-131)
# ---------------------------------------------------------------------------------------------------------------------
class _GradientPenalty(Function):
    """gp = mean((|| d D(x_hat) / d x_hat ||_2 - 1)^2), x_hat = alpha * real + (1 - alpha) * fake, differentiable with respect to
    D's parameters: forward and backward are one PHASE_GP_ONLY call of a step engine that holds D's weights - the forward pass,
    the first backward to the image, and (CGAN) the closed-form double backward through conv, train-mode BatchNorm, LeakyReLU,
    Linear, Dropout and the sigmoid (derivation checked against autograd(create_graph=True) in
    tests/test_gp_double_backward_math.py).  DCGAN never back-propagates its penalty (train/dcgan_trainer.py:178-179 only logs
    it): its engine forms the value and the Function returns zero gradients."""

    @staticmethod
    def forward(ctx, eng, real, fake, alpha, labels, mask, names, *params):
        views = eng.named_views("d")
        with torch.no_grad():
            for k, p in zip(names, params):                     # the engine computes with the module's CURRENT weights
                if views[k].data_ptr() != p.data_ptr():
                    views[k].copy_(p.detach().to(views[k].dtype).view_as(views[k]))
            eng.mark_weights_changed()
            norms = eng.gradient_penalty_pass(real, fake, alpha, labels, mask)
            gp = ((norms - 1.0) ** 2).mean()
            gviews = eng.named_views("d", "grads")
            ctx.grads = [gviews[k].detach().clone().view_as(p) for k, p in zip(names, params)] if eng.family == 1 else None
            ctx.shapes = [p.shape for p in params]
        return gp.clone()

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        if ctx.grads is None:
            raise JckError("the DCGAN penalty is a logged value only (train/dcgan_trainer.py:178-179): it has no gradient path")
        return (None,) * 7 + tuple(g * t for t in ctx.grads)


def gradient_penalty(model_d, real_data, fake_data, labels=None, alpha=None, drop_mask=None, prec=None, engine=None):
    """`compute_gradient_penalty` of the reference's trainers for the HIP modules, as a tensor that can be back-propagated
    (CGAN: `error_d = error_real + error_fake + 10 * gp; error_d.backward()`, train/cgan_trainer.py:200-203).
    model_d: model.CGAN.Discriminator (labels required) or model.DCGAN.Discriminator of this package.  alpha ~ U[0,1) [B,1,1,1]
    and the Dropout keep-mask (CGAN, p = 0.25) are drawn here when not given.  engine: a step engine to run on (default: one per
    (module, batch), cached on the module; its D arena is overwritten with the module's weights on every call).
    BatchNorm running statistics of the module are not moved by this pass (the reference's D(interpolates) call moves them once)."""
    from .engine import CganEngine, DcganEngine
    _need_cuda(real_data, "gradient_penalty")
    B = real_data.size(0)
    cg = labels is not None
    p_ = prec or getattr(model_d, "prec", None) or __import__("os").environ.get("JCKGAN_PREC", "bf16")
    if engine is None:
        cache = model_d.__dict__.setdefault("_jck_gp_engines", {})
        key = (B, p_, real_data.device.index)
        if key not in cache:
            size = getattr(model_d, "image_size", 64)
            cache[key] = (CganEngine(batch=B, prec=p_, device=real_data.device) if cg else
                          DcganEngine(batch=B, prec=p_, device=real_data.device, **({"image_size": size} if size != 64 else {})))
        engine = cache[key]
    if alpha is None:
        alpha = torch.rand(B, 1, 1, 1, device=real_data.device)
    if cg and drop_mask is None:
        drop_mask = (torch.rand(B, 256, device=real_data.device) >= 0.25).float()
    names = [k for k, _ in model_d.named_parameters()]
    params = [p for _, p in model_d.named_parameters()]
    with torch.no_grad():                                       # buffers (running statistics) ride along with the weights
        views = engine.named_views("d")
        for k, b in model_d.named_buffers():
            if k in views and views[k].data_ptr() != b.data_ptr():
                views[k].copy_(b.detach().to(views[k].dtype).view_as(views[k]))
    return _GradientPenalty.apply(engine, real_data, fake_data, alpha.reshape(B), labels, drop_mask, names, *params)
