"""Helpers for the GPU parity tests: NCHW fp32 (CPU, PyTorch layout) <-> NHWC T (device, library layout)."""
import torch

from hipgan import PREC_BF16, PREC_F32, lib  # noqa: F401
from hipgan._lib import cur_stream  # noqa: F401

DT = {PREC_BF16: torch.bfloat16, PREC_F32: torch.float32}
# relative-to-max tolerances per precision: F32 = exact fp32 MFMA (summation order only), BF16 = bf16 storage (2^-9)
TOL = {PREC_F32: 3e-6, PREC_BF16: 1.5e-2}


def to_nhwc(x_nchw, prec, cpad=None):
    """CPU NCHW fp32 -> device NHWC of the library's element type, channels zero-padded to cpad."""
    n, c, h, w = x_nchw.shape
    cp = cpad or (4 if c == 3 else c)
    t = torch.zeros(n, h, w, cp, dtype=torch.float32)
    t[..., :c] = x_nchw.permute(0, 2, 3, 1)
    return t.to(DT[prec]).cuda().contiguous()


def from_nhwc(t, c=None):
    """device NHWC -> CPU NCHW fp32 (first c channels)."""
    t = t.float().cpu()
    if c is not None:
        t = t[..., :c]
    return t.permute(0, 3, 1, 2).contiguous()


def rnd(x, prec):
    """Round a CPU fp32 tensor the way the library stores it (bf16 in fast mode)."""
    return x.to(torch.bfloat16).float() if prec == PREC_BF16 else x


def pack_down(w, prec):
    cs, cb = w.shape[0], w.shape[1]
    wp = torch.empty(lib.jck_pad_rows(cs) * 16 * lib.jck_pad_chan(cb), dtype=DT[prec], device="cuda")
    lib.jck_pack_down(prec, w.cuda().contiguous(), cs, cb, wp, cur_stream())
    return wp


def pack_up(w, prec):
    cs, cb = w.shape[0], w.shape[1]
    wp = torch.empty(4 * lib.jck_pad_rows(cb) * 4 * cs, dtype=DT[prec], device="cuda")
    lib.jck_pack_up(prec, w.cuda().contiguous(), cs, cb, wp, cur_stream())
    return wp


def stats_buf(pixels, c, nyrep=1):
    import ctypes
    return torch.full((lib.jck_stats_floats(pixels, c, nyrep),), float("nan"), device="cuda"), ctypes.c_int(0)


def stats_sum(buf, slots, c):
    """[slots][2][C] partials -> (sum[C], sumsq[C]) on the CPU."""
    v = buf[:slots.value * 2 * c].view(slots.value, 2, c).double().sum(0).float().cpu()
    return v[0], v[1]


def check(got, ref, tol, what=""):
    got, ref = got.double(), ref.double()
    assert got.shape == ref.shape, f"{what}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = ref.abs().max().item() + 1e-30
    err = (got - ref).abs()
    m = err.max().item()
    if not (m <= tol * scale):
        idx = torch.nonzero(err == err.max())[0].tolist()
        nbad = int((err > tol * scale).sum())
        raise AssertionError(f"{what}: max err {m:.3e} > {tol:.1e} * {scale:.3e}; {nbad}/{err.numel()} bad; first worst at {idx}: "
                             f"got {got[tuple(idx)].item():.6g} ref {ref[tuple(idx)].item():.6g}; "
                             f"rel-l2 {((got-ref).norm()/ref.norm()).item():.3e}")
    return m / scale
