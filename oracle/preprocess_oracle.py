"""CPU restatement of the reference's input transform (TEST INFRASTRUCTURE ONLY - never imported by the product path).

preprocess/dcgan_data_preprocessor.py:38-43 composes, per CIFAR image (a 32x32 uint8 PIL image):
    transforms.Resize(64)  ->  transforms.ToTensor()  ->  transforms.Normalize((0.5,)*3, (0.5,)*3)
Resize on a PIL image calls PIL.Image.resize(size, BILINEAR).  The arithmetic lives in Pillow (third-party, not vendored
in /root/reference; the repo pins no version - Pillow 12.2.0 is what this image ships): ImagingResample runs a horizontal
pass and then a vertical pass, each accumulating 8-bit pixels against coefficients quantised to 22 fractional bits and
rounding the result to uint8.  For an exact 2x upscale the bilinear support is one source pixel either side, the interior
coefficients are 3/4 and 1/4 (exact in 22 bits) and the border windows are clipped and renormalised to a single tap, so

    out[2k]   = (a[k-1] + 3 a[k] + 2) >> 2        out[2k+1] = (3 a[k] + a[k+1] + 2) >> 2        (indices clamped)

Pinned against Pillow itself by tests/golden/resize_u8.json (made by tests/golden/make_golden_resize.py)."""
import numpy as np


def _up2(x, axis):
    x = np.moveaxis(x.astype(np.int32), axis, -1)
    prev = np.concatenate([x[..., :1], x[..., :-1]], -1)
    nxt = np.concatenate([x[..., 1:], x[..., -1:]], -1)
    out = np.empty(x.shape[:-1] + (2 * x.shape[-1],), np.int32)
    out[..., 0::2] = (prev + 3 * x + 2) >> 2
    out[..., 1::2] = (3 * x + nxt + 2) >> 2
    return np.moveaxis(out, -1, axis)


def resize2x_u8(x):
    """uint8 [..., H, W] -> uint8 [..., 2H, 2W]; horizontal pass first, then vertical (Pillow's order)."""
    return _up2(_up2(x, -1), -2).astype(np.uint8)


def transform(x_u8):
    """uint8 [N,3,32,32] -> float32 [N,3,64,64] in [-1,1]: Resize(64), ToTensor (/255 in fp32), Normalize(0.5, 0.5)."""
    up = resize2x_u8(x_u8).astype(np.float32)
    return ((up / np.float32(255.0)) - np.float32(0.5)) / np.float32(0.5)
