"""DCGAN generator / discriminator for MI355X - drop-in for the reference's `model/DCGAN.py`.

Same public surface (reference model/DCGAN.py:6-76): no-arg `Generator()` / `Discriminator()`, sub-module names
`conv1..5`, `norm1..4` (identical `state_dict()` keys and default initialisation, because the parameter containers
are the very same torch.nn layer classes), `forward(x)`, and `weights_init(m)` for `.apply`.

One optional constructor keyword, `image_size` (default 64 = the reference): 128 builds the BASELINE.json configs[4]
topology - one more stride-2 stage at the deep end (D 3-64-128-256-512-1024-1, G 100-1024-512-256-128-64-3; modules
`conv1..6`, `norm1..5`).  The reference has no 128x128 behaviour (its nets are hard-wired to 3x64x64), so that plan is the
build's own; its oracle is the CPU restatement in oracle/gan_oracle.py (parity unpinned against the reference).

What differs is what runs: `forward` never calls ATen convolution / batch-norm kernels.  It hands NHWC tensors to
hand-written gfx950 kernels through the C ABI (`hipgan.functional`): MFMA gather-GEMMs for Conv2d /
ConvTranspose2d, fused BatchNorm statistics in the GEMM epilogue, vectorised normalise+activation passes.  There is
no CPU fallback: a CPU tensor raises.  Training uses the native step engine (train/dcgan_trainer.py) which shares
these modules' parameters zero-copy.
"""
import os

import torch
from torch import nn


def _prec():
    return os.environ.get("JCKGAN_PREC", "bf16")


class Discriminator(nn.Module):
    """[B,3,64,64] -> [B,1,1,1] probability.  4 x (Conv k4 s2 p1 -> BN -> LeakyReLU 0.2), Conv k4 s1 p0, sigmoid."""

    CHANNELS = (3, 64, 128, 256, 512)

    def __init__(self, image_size=64):
        super().__init__()
        if image_size not in (64, 128):
            raise ValueError("image_size must be 64 (the reference) or 128")
        c = self.CHANNELS + ((1024,) if image_size == 128 else ())
        n = len(c) - 1
        self.image_size = image_size
        for i in range(1, n + 1):
            setattr(self, f"conv{i}", nn.Conv2d(c[i - 1], c[i], kernel_size=4, stride=2, padding=1, bias=False))
            setattr(self, f"norm{i}", nn.BatchNorm2d(c[i]))
            setattr(self, f"relu{i}", nn.LeakyReLU(0.2, inplace=True))
        setattr(self, f"conv{n + 1}", nn.Conv2d(c[n], 1, kernel_size=4, stride=1, padding=0, bias=False))
        self.sigmoid = nn.Sigmoid()
        self.prec = None        # None -> $JCKGAN_PREC or "bf16"

    def forward(self, x):
        from hipgan import functional as HF
        return HF.dcgan_discriminator(self, x, self.prec or _prec())


class Generator(nn.Module):
    """[B,100,1,1] -> [B,3,64,64] in (-1,1).  ConvT k4 s1 p0, 3 x ConvT k4 s2 p1 (each BN + ReLU), ConvT k4 s2 p1, tanh."""

    CHANNELS = (100, 512, 256, 128, 64, 3)

    def __init__(self, image_size=64):
        super().__init__()
        if image_size not in (64, 128):
            raise ValueError("image_size must be 64 (the reference) or 128")
        c = self.CHANNELS if image_size == 64 else (100, 1024) + self.CHANNELS[1:]
        n = len(c) - 2                                     # BatchNorm stages
        self.image_size = image_size
        self.conv1 = nn.ConvTranspose2d(c[0], c[1], kernel_size=4, stride=1, padding=0, bias=False)
        self.norm1 = nn.BatchNorm2d(c[1])
        self.relu1 = nn.ReLU(inplace=True)
        for i in range(2, n + 1):
            setattr(self, f"conv{i}", nn.ConvTranspose2d(c[i - 1], c[i], kernel_size=4, stride=2, padding=1, bias=False))
            setattr(self, f"norm{i}", nn.BatchNorm2d(c[i]))
            setattr(self, f"relu{i}", nn.ReLU(inplace=True))
        setattr(self, f"conv{n + 1}", nn.ConvTranspose2d(c[n], c[n + 1], kernel_size=4, stride=2, padding=1, bias=False))
        self.tanh = nn.Tanh()
        self.prec = None

    def forward(self, x):
        from hipgan import functional as HF
        return HF.dcgan_generator(self, x, self.prec or _prec())


def weights_init(m):
    """Conv* weights ~ N(0, 0.02); BatchNorm weight ~ N(1, 0.02), bias 0 - matched by class name like the reference."""
    kind = type(m).__name__
    if "Conv" in kind:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif "BatchNorm" in kind:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)
