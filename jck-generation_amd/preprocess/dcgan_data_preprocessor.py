"""Input pipeline with the reference's interface (preprocess/dcgan_data_preprocessor.py): `DCGANDataPreprocessor(args)`,
`transform_data()`, `get_data_loader() -> (train_loader, metric_source)`.

The reference downloads CIFAR-100 through torchvision (no network / no torchvision here).  This version reads a LOCAL
copy of the CIFAR-100 python pickle (`./data/cifar-100-python/train`, the very file torchvision unpacks) when present and
otherwise falls back to seeded synthetic images of the same shape and range - the training hot path is identical.  Resize
32->64 (bilinear, as transforms.Resize(64) on a PIL image approximates) and Normalize(0.5, 0.5) run as tensor ops."""
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
        n = synthetic_size or 50000
        g = torch.Generator().manual_seed(2024)
        self._logger.warning(f"no local CIFAR-100 under {CIFAR_DIR}: using {n} seeded synthetic 32x32 images")
        x = (torch.rand(n, 3, 32, 32, generator=g) * 255).to(torch.uint8)
        return x, torch.randint(0, 100, (n,), generator=g).tolist()

    def transform_data(self):
        x = self.images.float() / 255.0                                           # ToTensor
        up = torch.nn.functional.interpolate(x, size=64, mode="bilinear", align_corners=False)    # Resize(64)
        self._train = (up - 0.5) / 0.5                                            # Normalize(0.5, 0.5)
        self._metric = _TensorSource(self.images, self.targets)                   # 299x299 resize is done lazily by metrics.py
        self._logger.debug("data transform")

    def get_data_loader(self):
        if self._train is None:
            self.transform_data()
        ds = torch.utils.data.TensorDataset(self._train)
        sampler = None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True)
        loader = torch.utils.data.DataLoader(ds, self.batch_size, shuffle=sampler is None, sampler=sampler,
                                             num_workers=self.num_worker, pin_memory=torch.cuda.is_available())
        return loader, self._metric
