"""Conditional DCGAN for MI355X - drop-in for the reference's `model/CGAN.py` (the live classes at :79-171).

Same surface: no-arg `Generator()` / `Discriminator()`, `forward(x, labels)` with one-hot int64 labels [B,100], sub-module
names `conv1..5`, `norm1..4`, `label_embedding`, `linear1`, `linear2` (identical state-dict keys and default init - the
containers are the same torch.nn classes, so `weights_init` leaves the Linear layers at their default init exactly like the
reference), `weights_init(m)`.  `forward` runs on the gfx950 kernels (hipgan.functional); there is no CPU fallback.
Training uses the native step engine (train/cgan_trainer.py), including the back-propagated gradient penalty."""
import os

import torch
from torch import nn


def _prec():
    return os.environ.get("JCKGAN_PREC", "bf16")


class Discriminator(nn.Module):
    """[B,3,64,64], one-hot [B,100] -> [B,1].  Label MLP (Linear(100,200)+LeakyReLU), 4 x (Conv k4 s2 p1 -> BN -> LeakyReLU),
    flatten, concat, Linear(8392,256), Dropout(0.25), Linear(256,1), sigmoid."""

    CHANNELS = (3, 64, 128, 256, 512)

    def __init__(self):
        super().__init__()
        self.label_embedding = nn.Linear(100, 200)
        self.label_embedding_relu1 = nn.LeakyReLU(0.2, inplace=True)
        c = self.CHANNELS
        for i in range(1, 5):
            setattr(self, f"conv{i}", nn.Conv2d(c[i - 1], c[i], kernel_size=4, stride=2, padding=1, bias=False))
            setattr(self, f"norm{i}", nn.BatchNorm2d(c[i]))
            setattr(self, f"relu{i}", nn.LeakyReLU(0.2, inplace=True))
        self.flatten = nn.Flatten()
        self.linear1 = nn.Linear(8192 + 200, 256)
        self.drop1 = nn.Dropout(0.25)
        self.linear2 = nn.Linear(256, 1)
        self.sigmoid = nn.Sigmoid()
        self.prec = None

    def forward(self, x, labels):
        from hipgan import functional as HF
        return HF.cgan_discriminator(self, x, labels, self.prec or _prec())


class Generator(nn.Module):
    """[B,100,1,1] noise + one-hot [B,100] -> [B,3,64,64].  cat -> ConvT(200->512, k4 s1 p0) then the DCGAN generator."""

    CHANNELS = (200, 512, 256, 128, 64, 3)

    def __init__(self):
        super().__init__()
        c = self.CHANNELS
        self.conv1 = nn.ConvTranspose2d(c[0], c[1], kernel_size=4, stride=1, padding=0, bias=False)
        self.norm1 = nn.BatchNorm2d(c[1])
        self.relu1 = nn.ReLU(inplace=True)
        for i in range(2, 5):
            setattr(self, f"conv{i}", nn.ConvTranspose2d(c[i - 1], c[i], kernel_size=4, stride=2, padding=1, bias=False))
            setattr(self, f"norm{i}", nn.BatchNorm2d(c[i]))
            setattr(self, f"relu{i}", nn.ReLU(inplace=True))
        self.conv5 = nn.ConvTranspose2d(c[4], c[5], kernel_size=4, stride=2, padding=1, bias=False)
        self.tanh = nn.Tanh()
        self.prec = None

    def forward(self, x, labels):
        from hipgan import functional as HF
        labels = labels.reshape(-1, 100, 1, 1)
        return HF.dcgan_generator(self, torch.cat([x, labels.to(x.dtype)], 1), self.prec or _prec())


def weights_init(m):
    """Conv* ~ N(0, 0.02); BatchNorm weight ~ N(1, 0.02), bias 0 (class-name match; Linear layers keep their default init)."""
    kind = type(m).__name__
    if "Conv" in kind:
        nn.init.normal_(m.weight.data, 0.0, 0.02)
    elif "BatchNorm" in kind:
        nn.init.normal_(m.weight.data, 1.0, 0.02)
        nn.init.constant_(m.bias.data, 0)
