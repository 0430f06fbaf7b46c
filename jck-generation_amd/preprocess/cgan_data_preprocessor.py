"""CGAN input pipeline with the reference's interface (preprocess/cgan_data_preprocessor.py): `CGANDataPreprocessor(args)`,
`transform_data()`, `get_data_loader() -> (train_loader of (image, one-hot int64 [100]) batches, metric source with
.targets)`, `idx_to_labels`.  Local CIFAR-100 pickle when present, seeded synthetic data otherwise (see the DCGAN
preprocessor for the rationale)."""
import os
import pickle

import torch

from preprocess.dcgan_data_preprocessor import CIFAR_DIR, DCGANDataPreprocessor, DeviceLoader, _TensorSource


class OneHotEncoder:
    """label -> LongTensor[label_count] one-hot (reference preprocess/cgan_data_preprocessor.py:11-16)."""

    def __init__(self, label_count):
        self.label_count = label_count

    def __call__(self, label):
        out = torch.zeros(self.label_count, dtype=torch.int64)
        out[int(label)] = 1
        return out


class CGANDataPreprocessor(DCGANDataPreprocessor):
    def __init__(self, args, synthetic_size=None):
        super().__init__(args, synthetic_size)
        self.idx_to_labels = {i: str(i) for i in range(100)}
        meta = os.path.join(CIFAR_DIR, "meta")
        if synthetic_size is None and os.path.exists(meta):
            with open(meta, "rb") as f:
                names = pickle.load(f, encoding="latin1")["fine_label_names"]
            self.idx_to_labels = {i: n for i, n in enumerate(names)}

    def get_data_loader(self):
        if self._train is None:
            self.transform_data()
        onehot = torch.nn.functional.one_hot(torch.tensor(self.targets, dtype=torch.int64), 100).to(torch.int64)
        if self._train.dtype == torch.uint8:                                       # device-resident dataset: labels live in HBM too
            return DeviceLoader(self._train, self.batch_size, onehot=onehot.to(self._train.device), seed=12345), self._metric
        ds = torch.utils.data.TensorDataset(self._train, onehot)
        sampler = None
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True)
        loader = torch.utils.data.DataLoader(ds, self.batch_size, shuffle=sampler is None, sampler=sampler,
                                             num_workers=self.num_worker, pin_memory=torch.cuda.is_available())
        return loader, self._metric
