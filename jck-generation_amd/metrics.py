"""Inception-Score / FID / intra-FID with the reference's interface (metrics.py:19-141): `Metrics(real_images)`,
`.inception_score(loader, splits=10)`, `.fid(loader, intra_fid=False, label=0)`, `.intra_fid(tensor)`.

The arithmetic is the reference's (100-d logits of a CIFAR-100 fine-tuned Inception-v3, softmax/KL per split, mean/cov +
scipy sqrtm, 20 superclass FIDs summed and divided by 100).  The feature extractor is pluggable; by default it is the
reference's network (inception_v3 with a Linear(2048,100) head, weights from ./save/iception_v3/loss_bset.pt, metrics.py:
46-51) run by the hand-written HIP chain of `inception.InceptionV3Hip` - no torchvision needed; when the weights file is
missing, construction raises MetricsUnavailable and the trainer carries on without scores.  Feature means and covariances are
formed in fp64 on the device (jck_mean_cov_f64) when the features live there; the matrix square root stays on the host
(scipy), as in the reference.  Fixes the reference's `.targets` defect for DCGAN (its loader has none): targets are optional."""
import os
import pickle

import numpy as np
import torch
from scipy.linalg import sqrtm

from utils import get_default_device

SUPERCLASS_MEMBERS = (
    (4, 30, 55, 72, 95), (1, 32, 67, 73, 91), (54, 62, 70, 82, 92), (9, 10, 16, 28, 61), (0, 51, 53, 57, 83),
    (22, 39, 40, 86, 87), (5, 20, 25, 84, 94), (6, 7, 14, 18, 24), (3, 42, 43, 88, 97), (12, 17, 37, 68, 76),
    (23, 33, 49, 60, 71), (15, 19, 21, 31, 38), (34, 63, 64, 66, 75), (26, 45, 77, 79, 99), (2, 11, 35, 46, 98),
    (27, 29, 44, 78, 93), (36, 50, 65, 74, 80), (47, 52, 56, 59, 96), (8, 13, 48, 58, 90), (41, 69, 81, 85, 89))


class MetricsUnavailable(RuntimeError):
    pass


def default_extractor(device, weights="./save/iception_v3/loss_bset.pt"):
    """The reference's metric network from a LOCAL weights file (torchvision key names), on the HIP kernels."""
    if not os.path.exists(weights):
        raise MetricsUnavailable(f"fine-tuned Inception weights not found at {weights} (the reference loads them at metrics.py:51; "
                                 f"they are not part of its repository)")
    if not torch.cuda.is_available():
        raise MetricsUnavailable("the metric network runs on the GPU (no CPU fallback)")
    from inception import InceptionV3Hip
    return InceptionV3Hip.from_file(weights, device)


def mean_cov_device(x):
    """(mean [D], covariance [D,D]) of a CUDA tensor as fp64 DEVICE tensors (jck_mean_cov_f64; no host sync)."""
    from hipgan._lib import cur_stream, lib
    x = x.to(torch.float32).contiguous()
    n, d = x.shape
    mu = torch.empty(d, dtype=torch.float64, device=x.device)
    cov = torch.empty(d, d, dtype=torch.float64, device=x.device)
    lib.jck_mean_cov_f64(x, mu, cov, n, d, cur_stream())
    return mu, cov


def mean_cov(x):
    """(mean [D], covariance [D,D]) in float64 as np.mean(axis=0) / np.cov(rowvar=False) give them (metrics.py:120-126).  A
    CUDA tensor is reduced on the device (fp64 accumulation, fixed summation order); anything else by numpy."""
    if torch.is_tensor(x) and x.is_cuda:
        mu, cov = mean_cov_device(x)
        return mu.cpu().numpy(), cov.cpu().numpy()
    x = x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)
    return np.mean(x, axis=0), np.cov(x, rowvar=False)


def fid_from_stats(mu1, sigma1, mu2, sigma2):
    """reference metrics.py:127-129 from the two Gaussians' parameters"""
    mu1, sigma1, mu2, sigma2 = (np.asarray(v, dtype=np.float64) for v in (mu1, sigma1, mu2, sigma2))
    covmean = sqrtm(sigma1.dot(sigma2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(np.sum((mu1 - mu2) ** 2.0) + np.trace(sigma1 + sigma2 - 2.0 * covmean))


def fid_from_features(real, fake):
    """Frechet distance between the Gaussians fitted to two feature matrices (reference metrics.py:120-129)."""
    (mu1, sigma1), (mu2, sigma2) = mean_cov(real), mean_cov(fake)
    covmean = sqrtm(sigma1.dot(sigma2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(np.sum((mu1 - mu2) ** 2.0) + np.trace(sigma1 + sigma2 - 2.0 * covmean))


def inception_score_from_probs(preds, splits=10):
    """exp(mean KL(p(y|x) || p(y))) per split, averaged (reference metrics.py:97-110; scipy.stats.entropy semantics)."""
    n = preds.shape[0]
    out = []
    for k in range(splits):
        part = preds[k * (n // splits):(k + 1) * (n // splits), :]
        py = np.mean(part, axis=0)
        pk = part / part.sum(axis=1, keepdims=True)
        qk = py / py.sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            kl = np.where(pk > 0, pk * np.log(pk / qk), 0.0).sum(axis=1)
        out.append(np.exp(np.mean(kl)))
    return float(np.mean(out))


class Metrics:
    def __init__(self, real_images, extractor=None, real_features=None, cache="./data/metric_data.pikl"):
        self.device = get_default_device()
        self.class_to_superclass = {c: s for s, row in enumerate(SUPERCLASS_MEMBERS) for c in row}
        self.inception_model = extractor if extractor is not None else (None if real_features is not None
                                                                        else default_extractor(self.device))
        real_targets = getattr(real_images, "targets", None)
        fake_targets = [i for i in range(100) for _ in range(10)]
        self.real_superclass_idx, self.fake_superclass_idx = {}, {}
        for s in range(20):
            self.fake_superclass_idx[s] = [i for i, t in enumerate(fake_targets) if self.class_to_superclass[t] == s]
            if real_targets is not None:
                self.real_superclass_idx[s] = [i for i, t in enumerate(real_targets) if self.class_to_superclass[int(t)] == s]
        if real_features is not None:
            self.real_features = np.asarray(real_features)
        elif os.path.exists(cache):
            with open(cache, "rb") as f:
                self.real_features = pickle.load(f)
        else:
            loader = torch.utils.data.DataLoader(real_images, 128, shuffle=False, num_workers=0)
            self.real_features = self._extract(loader, real=True)
            os.makedirs(os.path.dirname(cache), exist_ok=True)
            with open(cache, "wb") as f:
                pickle.dump(self.real_features, f, pickle.HIGHEST_PROTOCOL)

    def _prep_real(self, image):
        """real images arrive as uint8 [B,3,32,32]: Resize((299,299)) + ImageNet normalisation (reference preprocessor :44-47)."""
        x = image.float() / 255.0 if image.dtype == torch.uint8 else image
        if x.shape[-1] != 299:
            x = torch.nn.functional.interpolate(x, size=[299, 299], mode="bilinear", align_corners=False)
            mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
            std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
            x = (x - mean) / std
        return x

    def _extract(self, images, real=False, softmax=False, keep_on_device=False):
        feats = []
        for image in images:
            if real:
                image = self._prep_real(image[0])
            if self.inception_model is None:
                feature = image if torch.is_tensor(image) else torch.as_tensor(image)     # pre-computed features
            else:
                with torch.no_grad():
                    feature = self.inception_model(image.to(self.device))
            if softmax:
                feature = torch.nn.functional.softmax(feature.float(), dim=1)
            feats.append(feature.detach())
        feats = torch.cat([f if torch.is_tensor(f) else torch.as_tensor(f) for f in feats])
        return feats if (keep_on_device and feats.is_cuda) else feats.cpu().numpy()

    # ---- one forward pass, every score (the evaluation branch of the trainers: the reference runs the network once per score on
    # the same images - metrics.py:98,114 - which yields the same features each time)
    def logits(self, images, batch=128):
        """images: a tensor [N,3,299,299] already prepared for the network (device or host) -> logits [N,100] where the
        extractor left them."""
        out = []
        for i in range(0, images.shape[0], batch):
            chunk = images[i:i + batch]
            if self.inception_model is None:
                out.append(chunk)
                continue
            with torch.no_grad():
                out.append(self.inception_model(chunk.to(self.device)).detach())
        return torch.cat(out)

    def fake_stats_device(self, logits, intra=False):
        """Device part of the FID of generated logits (CUDA): {'mu', 'cov'[, 'mu_s<k>', 'cov_s<k>' per superclass]} as fp64 device
        tensors - no host sync, so it can sit on a side stream; finish with scores_from_stats()."""
        out = {}
        out["mu"], out["cov"] = mean_cov_device(logits)
        if intra:
            for s_ in range(20):
                idx = torch.as_tensor(self.fake_superclass_idx[s_], device=logits.device)
                out[f"mu_s{s_}"], out[f"cov_s{s_}"] = mean_cov_device(logits[idx])
        return out

    def _real_stats(self, label=None):
        """(mean, cov) of the cached real features (of one superclass), computed once - on the device when there is one"""
        cache = self.__dict__.setdefault("_real_stats_cache", {})
        if label not in cache:
            dev = self._real_on_device()
            if label is None:
                cache[label] = mean_cov(dev if dev is not None else self.real_features)
            else:
                idx = self.real_superclass_idx[label]
                cache[label] = mean_cov(dev[torch.as_tensor(idx, device=dev.device)] if dev is not None else self.real_features[idx])
        return cache[label]

    def scores_from_stats(self, logits, stats, splits=10, intra=False):
        """Host part: logits (host) -> inception score; stats (host copies of fake_stats_device) -> FID[, intra-FID]."""
        probs = torch.nn.functional.softmax(torch.as_tensor(logits).float(), dim=1).numpy()
        is_ = inception_score_from_probs(probs, splits)
        g = lambda v: v.numpy() if torch.is_tensor(v) else v
        fid = fid_from_stats(*self._real_stats(), g(stats["mu"]), g(stats["cov"]))
        if not intra:
            return is_, fid
        total = sum(fid_from_stats(*self._real_stats(s_), g(stats[f"mu_s{s_}"]), g(stats[f"cov_s{s_}"])) for s_ in range(20))
        return is_, fid, total / 100

    def scores_from_logits(self, logits, splits=10, intra=False):
        """-> (inception score, FID[, intra-FID]) from the logits of the generated images; the same arithmetic as
        inception_score() / fid() / intra_fid()."""
        probs = torch.nn.functional.softmax(logits.float(), dim=1).cpu().numpy()
        is_ = inception_score_from_probs(probs, splits)
        on_dev = logits.is_cuda and self._real_on_device() is not None
        feats = logits if on_dev else logits.detach().cpu().numpy()
        fid = fid_from_features(self._real_on_device() if on_dev else self.real_features, feats)
        if not intra:
            return is_, fid
        total = 0
        for s_ in range(20):
            sub = feats[torch.as_tensor(self.fake_superclass_idx[s_], device=logits.device)] if on_dev else feats[self.fake_superclass_idx[s_]]
            idx = self.real_superclass_idx[s_]
            real = self._real_on_device()[torch.as_tensor(idx, device=logits.device)] if on_dev else self.real_features[idx]
            total += fid_from_features(real, sub)
        return is_, fid, total / 100

    def inception_score(self, images, splits=10):
        return inception_score_from_probs(self._extract(images, softmax=True), splits)

    def _real_on_device(self):
        """the cached real features, uploaded once when a GPU is there (their mean / covariance are then formed on it)"""
        if not torch.cuda.is_available():
            return None
        if getattr(self, "_real_dev", None) is None:
            self._real_dev = torch.as_tensor(np.asarray(self.real_features, dtype=np.float32)).to(self.device)
        return self._real_dev

    def fid(self, generated_images, intra_fid=False, label=0):
        gen = self._extract(generated_images, keep_on_device=True)
        dev = self._real_on_device() if (torch.is_tensor(gen) and gen.is_cuda) else None
        if intra_fid:
            idx = self.real_superclass_idx[label]
            real = dev[torch.as_tensor(idx, device=dev.device)] if dev is not None else self.real_features[idx]
        else:
            real = dev if dev is not None else self.real_features
        return fid_from_features(real, gen)

    def intra_fid(self, generated_images):
        total = 0
        for s in range(20):
            sub = generated_images[self.fake_superclass_idx[s]]
            total += self.fid(torch.utils.data.DataLoader(sub, 128, shuffle=False), intra_fid=True, label=s)
        return total / 100
