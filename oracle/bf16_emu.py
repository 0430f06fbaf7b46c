"""bf16-storage emulation of the DCGAN nets on the CPU.  TEST INFRASTRUCTURE ONLY (same rule as gan_oracle.py).

The HIP fast path (JCK_PREC_BF16) keeps activations, activation gradients and the packed GEMM operands in bf16 and
accumulates in fp32.  A GAN step amplifies that rounding (5.8e-4 / 3.4e-3 on loss_d / loss_g at
step 0, 1e-1 by step 25), so "how far may the fast path be from the fp32 oracle" needs an envelope that is derived, not
guessed.  This file restates the reference's nets (model/DCGAN.py:29-35, :61-67) with a rounding to bf16 at exactly the
places where the HIP path stores a tensor in bf16:

    packed weights (conv / convT operands)          -> bf16, gradient stays fp32          (_RndW)
    images real_noisy / fake / x_hat (NHWC4)        -> bf16, their gradients too          (_Rnd)
    conv output y                                   -> bf16; BatchNorm statistics are taken from the fp32 accumulators
    a = act(scale*y + shift)                        -> bf16
    g_a (dgrad output), g_y (BatchNorm backward)    -> bf16
    tanh output, its backward                       -> bf16
    head (D.conv5): fp32 weights on bf16 a4; g_a4   -> bf16

`tests/test_bf16_envelope.py` uses it two ways: (1) the HIP bf16 step must agree with THIS restatement much more
closely than with the fp32 oracle (what is left is summation order), and (2) the distance emulation <-> fp32 oracle is the
envelope for the distance HIP <-> fp32 oracle, per step and along a free-running trajectory.

CGAN (round 4; model/CGAN.py:109-123,151-162, train/cgan_trainer.py:200-203): its penalty is BACK-PROPAGATED - a double backward
through conv, train-mode BatchNorm, LeakyReLU, Linear, Dropout, Sigmoid - which custom Functions with a hand-written first
backward do not reproduce.  The CGAN nets below are therefore written with plain differentiable ops and two primitives that stay
differentiable to any order: st(x) (forward value bf16(x), gradient the identity: straight-through rounding) for a tensor that is
STORED in bf16, and grad_st(x) (identity forward, st() applied to the gradient) for a tensor whose GRADIENT is stored in bf16.
The forward and first-backward storage points are those of the HIP path (conv outputs, activations, dgrad / BatchNorm-backward
outputs, the concat / hidden / dropped rows of the label head and their gradients, the three images and the penalty's image
gradient); the intermediates of the penalty's second-order sweep (the v-chain of csrc/engine.hip gp_double_backward, bf16 in
HBM on the device) are left in fp32 here - an envelope for the storage format, not a bit-level restatement of that sweep.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .gan_oracle import BN_EPS, BN_MOMENTUM, LRELU, _n_norm


def bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


class _Rnd(Function):
    """A tensor stored in bf16 whose gradient is stored in bf16 as well."""

    @staticmethod
    def forward(ctx, x):
        return bf(x)

    @staticmethod
    def backward(ctx, g):
        return bf(g)


class _RndW(Function):
    """A packed bf16 GEMM operand of an fp32 master weight: the weight gradient stays fp32."""

    @staticmethod
    def forward(ctx, w):
        return bf(w)

    @staticmethod
    def backward(ctx, g):
        return g


class _BnAct(Function):
    """BatchNorm2d (train) + (Leaky)ReLU exactly as the kernels compute it: statistics from the fp32 conv accumulators,
    y stored bf16, a = act(scale*y + shift) stored bf16; backward from the bf16 g_a and y with the saved mean / invstd:
    g_z = g_a*act'(z), g_y = scale*(g_z - mean g_z - xhat*mean(g_z xhat)) stored bf16 (csrc/ew.hpp: bn_finalize_kernel,
    bn_act_fwd_kernel, bn_bwd_reduce_kernel, bn_bwd_apply_kernel)."""

    @staticmethod
    def forward(ctx, y_acc, gamma, beta, slope):
        n = y_acc.numel() // y_acc.shape[1]
        yd = y_acc.double()
        mean_d = yd.mean((0, 2, 3))
        var_d = ((yd * yd).mean((0, 2, 3)) - mean_d * mean_d).clamp_min(0.0)
        mean, var = mean_d.float(), var_d.float()
        invstd = 1.0 / torch.sqrt(var + BN_EPS)
        sc = gamma * invstd
        sh = beta - mean * sc
        y_b = bf(y_acc)
        z = y_b * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        a = bf(torch.where(z > 0, z, slope * z))
        ctx.save_for_backward(y_b, sc, sh, mean, invstd)
        ctx.slope, ctx.n = slope, n
        unbiased = var * (n / max(n - 1.0, 1.0))
        ctx.mark_non_differentiable(mean, unbiased)
        return a, mean, unbiased

    @staticmethod
    def backward(ctx, g_a, _gm, _gv):
        y_b, sc, sh, mean, invstd = ctx.saved_tensors
        v = lambda t: t.view(1, -1, 1, 1)
        g_a = bf(g_a)                                     # the dgrad kernel stored it in bf16
        z = y_b * v(sc) + v(sh)
        gz = torch.where(z > 0, g_a, ctx.slope * g_a)
        xh = (y_b - v(mean)) * v(invstd)
        s1 = gz.sum((0, 2, 3))
        s2 = (gz * xh).sum((0, 2, 3))
        g_y = bf(v(sc) * (gz - v(s1) / ctx.n - xh * (v(s2) / ctx.n)))
        return g_y, s2, s1, None


def _bn_act(p, name, y_acc, slope, update_stats=True):
    a, mean, unbiased = _BnAct.apply(y_acc, p[f"{name}.weight"], p[f"{name}.bias"], slope)
    if update_stats:
        with torch.no_grad():
            p[f"{name}.num_batches_tracked"] += 1
            p[f"{name}.running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean)
            p[f"{name}.running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * unbiased)
    return a


class _TanhStore(Function):
    """fake_raw = tanh(acc) stored bf16; backward g*(1 - y^2) from the stored y, stored bf16 (tanh_bwd_kernel)."""

    @staticmethod
    def forward(ctx, acc):
        y = bf(torch.tanh(acc))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        return bf(g * (1 - y * y))


def store(x):
    """An image-side tensor as the HIP path keeps it (bf16 NHWC4)."""
    return _Rnd.apply(x)


def generator_bf16(p, z, labels=None):
    assert labels is None, "bf16 emulation covers DCGAN only"
    n = _n_norm(p)
    h = F.conv_transpose2d(bf(z), _RndW.apply(p["conv1.weight"]), None, 1, 0)
    h = _bn_act(p, "norm1", h, 0.0)
    for i in range(2, n + 1):
        h = F.conv_transpose2d(h, _RndW.apply(p[f"conv{i}.weight"]), None, 2, 1)
        h = _bn_act(p, f"norm{i}", h, 0.0)
    return _TanhStore.apply(F.conv_transpose2d(h, _RndW.apply(p[f"conv{n + 1}.weight"]), None, 2, 1))


def discriminator_bf16(p, x, labels=None, drop_mask=None, drop_p=0.25):
    """x must already be a stored (bf16-rounded) image: callers pass store(...)."""
    n = _n_norm(p)
    assert f"conv{n + 1}.weight" in p, "bf16 emulation covers DCGAN only"
    h = x
    for i in range(1, n + 1):
        h = F.conv2d(h, _RndW.apply(p[f"conv{i}.weight"]), None, 2, 1)
        h = _bn_act(p, f"norm{i}", h, LRELU)
    # head: fp32 weights on the bf16 a4; the gradient w.r.t. a4 is rounded by _BnAct.backward of the last layer
    return torch.sigmoid(F.conv2d(h, p[f"conv{n + 1}.weight"], None, 1, 0))


# ------------------------------------------------------------------------------------------------------------------------------
# CGAN: the same storage points with primitives that autograd can differentiate twice (the back-propagated penalty)
# ------------------------------------------------------------------------------------------------------------------------------
def st(x: torch.Tensor) -> torch.Tensor:
    """straight-through rounding: the value stored in bf16, the gradient passed through unchanged - differentiable to any order"""
    return x + (bf(x) - x).detach()


class _GradSt(Function):
    """identity whose gradient is stored in bf16 (st() on the way back, so a second backward differentiates through it)"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return st(g)


def grad_st(x):
    return _GradSt.apply(x)


def store_dd(x):
    """an image-side tensor kept in bf16 together with its gradient (real_noisy / fake / x_hat), twice differentiable"""
    return grad_st(st(x))


def _bn_act_dd(p, name, y_acc, slope, update_stats=True):
    """_BnAct with plain ops: statistics from the fp32 accumulators, y and a stored bf16, g_a and g_y stored bf16"""
    y_in = grad_st(y_acc)                                       # g_y (BatchNorm backward output) is stored in bf16
    n = y_in.numel() // y_in.shape[1]
    yd = y_in.double()
    mean_d = yd.mean((0, 2, 3))
    var_d = ((yd * yd).mean((0, 2, 3)) - mean_d * mean_d).clamp_min(0.0)
    mean, var = mean_d.float(), var_d.float()
    invstd = 1.0 / torch.sqrt(var + BN_EPS)
    sc = p[f"{name}.weight"] * invstd
    sh = p[f"{name}.bias"] - mean * sc
    z = st(y_in) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    a = grad_st(st(torch.where(z > 0, z, slope * z)))           # a stored bf16; g_a (the dgrad output) stored bf16
    if update_stats:
        with torch.no_grad():
            p[f"{name}.num_batches_tracked"] += 1
            p[f"{name}.running_mean"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            p[f"{name}.running_var"].mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * (var * (n / max(n - 1.0, 1.0))).detach())
    return a


def generator_cgan_bf16(p, z, labels=None):
    """model/CGAN.py:151-162: [z | one-hot] written in bf16 into the operand rows of conv1 (cgan_z_kernel), then the DCGAN stack"""
    assert labels is not None
    n = _n_norm(p)
    h = st(torch.cat([z, labels.reshape(-1, 100, 1, 1).float()], 1))
    h = F.conv_transpose2d(h, st(p["conv1.weight"]), None, 1, 0)
    h = _bn_act_dd(p, "norm1", h, 0.0)
    for i in range(2, n + 1):
        h = F.conv_transpose2d(h, st(p[f"conv{i}.weight"]), None, 2, 1)
        h = _bn_act_dd(p, f"norm{i}", h, 0.0)
    acc = grad_st(F.conv_transpose2d(h, st(p[f"conv{n + 1}.weight"]), None, 2, 1))      # tanh backward's output is stored bf16
    return st(torch.tanh(acc))                                                         # fake_raw stored bf16


def discriminator_cgan_bf16(p, x, labels=None, drop_mask=None, drop_p=0.25):
    """model/CGAN.py:109-123 with the engine's storage (csrc/engine.hip cg_head_forward / d_head_backward): conv stack as DCGAN;
    the label embedding is computed in fp32 and stored bf16 beside flatten(a4) in the concat rows; Linear(8392, 256) on packed
    bf16 weights with fp32 accumulation, its output and the dropped rows stored bf16 (linear_finish_kernel: both from the fp32
    sum); Linear(256, 1) + sigmoid on fp32 weights; the gradients of the dropped rows, the hidden rows and the concat rows are
    stored bf16.  x must already be a stored image (store_dd)."""
    assert labels is not None and drop_mask is not None
    n = _n_norm(p)
    h = x
    for i in range(1, n + 1):
        h = F.conv2d(h, st(p[f"conv{i}.weight"]), None, 2, 1)
        h = _bn_act_dd(p, f"norm{i}", h, LRELU)
    e = F.leaky_relu(F.linear(labels.float(), p["label_embedding.weight"], p["label_embedding.bias"]), LRELU)
    con = grad_st(torch.cat([h.flatten(1), st(e)], 1))
    s = F.linear(con, st(p["linear1.weight"]), p["linear1.bias"])
    s = grad_st(s)                                              # g_h stored bf16
    hd = grad_st(st(s * drop_mask / (1.0 - drop_p)))            # h_drop stored bf16 (from the fp32 sum); g_hd stored bf16
    return torch.sigmoid(F.linear(hd, p["linear2.weight"], p["linear2.bias"]))
