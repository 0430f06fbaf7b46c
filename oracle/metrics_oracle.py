"""CPU oracle for the IS / FID / intra-FID arithmetic of the reference's `metrics.py`.
TEST INFRASTRUCTURE ONLY (see oracle/gan_oracle.py header for who may import it).

Restates, with numpy/scipy, what metrics.py:97-141 computes from 100-d Inception logits.  The
feature extractor itself (fine-tuned Inception-v3, metrics.py:46-51) is "parity unpinned": its
weights are not available offline.  Pinned by tests/golden/metrics.json.
"""
import numpy as np
from scipy.linalg import sqrtm

# metrics.py:23-44 CIFAR-100 fine class -> superclass, restated as 20 rows of 5
SUPERCLASS_ROWS = [
    [4, 30, 55, 72, 95], [1, 32, 67, 73, 91], [54, 62, 70, 82, 92], [9, 10, 16, 28, 61], [0, 51, 53, 57, 83],
    [22, 39, 40, 86, 87], [5, 20, 25, 84, 94], [6, 7, 14, 18, 24], [3, 42, 43, 88, 97], [12, 17, 37, 68, 76],
    [23, 33, 49, 60, 71], [15, 19, 21, 31, 38], [34, 63, 64, 66, 75], [26, 45, 77, 79, 99], [2, 11, 35, 46, 98],
    [27, 29, 44, 78, 93], [36, 50, 65, 74, 80], [47, 52, 56, 59, 96], [8, 13, 48, 58, 90], [41, 69, 81, 85, 89]]


def class_to_superclass():
    return {c: s for s, row in enumerate(SUPERCLASS_ROWS) for c in row}


def softmax(x):
    x = x - x.max(axis=1, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=1, keepdims=True)


def inception_score(logits, splits=10):
    """metrics.py:97-110: softmax in fp32 (torch), KL(p(y|x) || p(y)) per split, exp(mean), mean over splits."""
    preds = softmax(logits.astype(np.float32)).astype(np.float32)
    n = preds.shape[0]
    scores = []
    for k in range(splits):
        part = preds[k * (n // splits):(k + 1) * (n // splits)]
        py = part.mean(axis=0)
        # scipy.stats.entropy(pk, qk) normalises both and returns sum(pk * log(pk / qk))
        kl = []
        for row in part:
            pk = row / row.sum()
            qk = py / py.sum()
            kl.append(float(np.sum(np.where(pk > 0, pk * np.log(pk / qk), 0.0))))
        scores.append(np.exp(np.mean(kl)))
    return float(np.mean(scores))


def fid(real_feats, fake_feats):
    """metrics.py:113-129."""
    mu1, s1 = real_feats.mean(axis=0), np.cov(real_feats, rowvar=False)
    mu2, s2 = fake_feats.mean(axis=0), np.cov(fake_feats, rowvar=False)
    diff = np.sum((mu1 - mu2) ** 2.0)
    covmean = sqrtm(s1.dot(s2))
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(diff + np.trace(s1 + s2 - 2.0 * covmean))


def intra_fid(real_feats, real_targets, fake_feats, fake_targets):
    """metrics.py:132-141: 20 superclass FIDs summed and divided by 100 (sic)."""
    c2s = class_to_superclass()
    rs = np.array([c2s[int(t)] for t in real_targets])
    fs = np.array([c2s[int(t)] for t in fake_targets])
    total = 0.0
    for s in range(20):
        total += fid(real_feats[rs == s], fake_feats[fs == s])
    return total / 100
