"""Input pipeline with the reference's interface (preprocess/dcgan_data_preprocessor.py): `DCGANDataPreprocessor(args)`,
`transform_data()`, `get_data_loader() -> (train_loader, metric_source)`.

The reference downloads CIFAR-100 through torchvision (no network / no torchvision here).  This version reads a LOCAL
copy of the CIFAR-100 python pickle (`./data/cifar-100-python/train`, the very file torchvision unpacks) when present and
otherwise falls back to seeded synthetic images of the same shape and range - the training hot path is identical.

MI355X-first layout: the uint8 dataset (150 MB for CIFAR) is copied to HBM ONCE; a batch is a vector of indices
(`hipgan.engine.DeviceBatch`) and the step itself gathers it and applies Resize(64) / ToTensor / Normalize(0.5, 0.5)
(reference :38-43) in the kernel that also adds the instance noise - no per-step host->device image traffic, no DataLoader
workers.  The transform is bit-exact against the reference's: transforms.Resize on a PIL image is PIL's bilinear resize,
whose 2x upscale is a horizontal and a vertical 3:1 pass each rounded to uint8 (`resize2x_pil_u8`, pinned to PIL by
tests/golden/resize_u8.json).  Without a GPU (and with JCKGAN_DEVICE_DATA=0) the same arithmetic runs as tensor ops on the
host and a torch DataLoader is returned, as in the reference."""
import os
import pickle

import numpy as np
import torch

from logger.main_logger import MainLogger

CIFAR_DIR = os.path.join(".", "data", "cifar-100-python")


class _TensorSource:
    """metric source: tensor dataset with `.targets` (what metrics.Metrics reads, reference metrics.py:56)."""

    def __init__(self, images, targets):
        self.images, self.targets = images, targets

    def __len__(self):
        return self.images.shape[0]

    def __getitem__(self, i):
        return self.images[i], self.targets[i]


def resize2x_pil_u8(x):
    """uint8 [..., H, W] -> uint8 [..., 2H, 2W] exactly as PIL.Image.resize(BILINEAR) upsamples by 2 (what
    transforms.Resize(64) does to a 32x32 PIL image, reference :39): horizontal pass, then vertical pass, each rounded to
    uint8; interior taps weigh 3:1, the clamped border taps reduce to a copy."""
    def up(t, dim):
        t = t.to(torch.int32).movedim(dim, -1)
        n = t.shape[-1]
        prev = torch.cat([t[..., :1], t[..., :-1]], -1)
        nxt = torch.cat([t[..., 1:], t[..., -1:]], -1)
        out = torch.empty(t.shape[:-1] + (2 * n,), dtype=torch.int32)
        out[..., 0::2] = (prev + 3 * t + 2) >> 2
        out[..., 1::2] = (3 * t + nxt + 2) >> 2
        return out.movedim(-1, dim)
    return up(up(x, -1), -2).to(torch.uint8)


class DeviceLoader:
    """Iterable with the DataLoader surface the trainers use (`len()`, iteration over `[images, (labels)]` batches): images
    are `DeviceBatch`es over the uint8 dataset in HBM, labels (CGAN: int64 one-hot [B,100]) are gathered on the device.
    Shuffles every epoch like DataLoader(shuffle=True); under torch.distributed every rank takes a strided share of the same
    permutation (DistributedSampler semantics, padded by wrap-around)."""

    def __init__(self, data_u8, batch_size, onehot=None, seed=0):
        from hipgan.engine import DeviceBatch
        self._DeviceBatch = DeviceBatch
        self.data, self.onehot, self.batch_size = data_u8, onehot, batch_size
        self.rank, self.world = 0, 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.rank, self.world = torch.distributed.get_rank(), torch.distributed.get_world_size()
        self.n_local = (data_u8.shape[0] + self.world - 1) // self.world
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (self.n_local + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        n = self.data.shape[0]
        perm = torch.randperm(n, generator=self.gen)
        if self.world > 1:
            total = self.n_local * self.world
            perm = torch.cat([perm, perm[:total - n]])[self.rank:total:self.world]
        perm = perm.to(self.data.device)
        for i in range(0, perm.numel(), self.batch_size):
            idx = perm[i:i + self.batch_size]
            batch = [self._DeviceBatch(self.data, idx)]
            if self.onehot is not None:
                batch.append(self.onehot[idx])
            yield batch


class DCGANDataPreprocessor:
    def __init__(self, args, synthetic_size=None):
        self._logger = MainLogger(args)
        self.batch_size = args.batch_size
        self.num_worker = args.num_worker
        self.images, self.targets = self._load(synthetic_size)
        self._train, self._metric = None, None
        self._logger.debug("data preprocessor init")

    def _load(self, synthetic_size):
        path = os.path.join(CIFAR_DIR, "train")
        if synthetic_size is None and os.path.exists(path):
            with open(path, "rb") as f:
                d = pickle.load(f, encoding="latin1")
            x = torch.from_numpy(np.asarray(d["data"], dtype=np.uint8).reshape(-1, 3, 32, 32))
            self._logger.debug(f"CIFAR-100 loaded from {path}: {tuple(x.shape)}")
            return x, [int(t) for t in d["fine_labels"]]
        if synthetic_size is None and os.environ.get("JCKGAN_SYNTHETIC", "0") != "1":
            # a training run on random pixels must be asked for: silently writing checkpoints of a model trained on noise
            # because a path was mistyped helps nobody (the reference downloads CIFAR-100; there is no network here)
            raise FileNotFoundError(f"no local CIFAR-100 under {CIFAR_DIR} (expected the pickle `train` of cifar-100-python); "
                                    f"set JCKGAN_SYNTHETIC=1 to train on seeded synthetic 32x32 images instead")
        n = synthetic_size or 50000
        g = torch.Generator().manual_seed(2024)
        self._logger.warning(f"no local CIFAR-100 under {CIFAR_DIR}: using {n} seeded synthetic 32x32 images")
        x = (torch.rand(n, 3, 32, 32, generator=g) * 255).to(torch.uint8)
        return x, torch.randint(0, 100, (n,), generator=g).tolist()

    @staticmethod
    def device_resident():
        return torch.cuda.is_available() and os.environ.get("JCKGAN_DEVICE_DATA", "1") != "0"

    def transform_data(self):
        self._metric = _TensorSource(self.images, self.targets)                   # 299x299 resize is done lazily by metrics.py
        if self.device_resident():
            self._train = self.images.cuda().contiguous()                         # uint8, transformed inside the step
        else:
            up = resize2x_pil_u8(self.images)                                     # Resize(64) on the PIL image (:39)
            self._train = (up.float() / 255.0 - 0.5) / 0.5                        # ToTensor, Normalize(0.5, 0.5)
        self._logger.debug("data transform")

    def get_data_loader(self):
        if self._train is None:
            self.transform_data()
        if self._train.dtype == torch.uint8:
            return DeviceLoader(self._train, self.batch_size, seed=12345), self._metric
        ds = torch.utils.data.TensorDataset(self._train)
        sampler = None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True)
        loader = torch.utils.data.DataLoader(ds, self.batch_size, shuffle=sampler is None, sampler=sampler,
                                             num_workers=self.num_worker, pin_memory=torch.cuda.is_available())
        return loader, self._metric
