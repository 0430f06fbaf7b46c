"""Shared helpers for the test-suite (digest comparison against tests/golden/*.json)."""
import hashlib
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def synth_images(n, seed=2024):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n, 3, 64, 64, generator=g) * 2 - 1


def synth_onehot(n, seed=2025, classes=100):
    g = torch.Generator().manual_seed(seed)
    lab = torch.randint(0, classes, (n,), generator=g)
    return torch.nn.functional.one_hot(lab, 100).to(torch.int64), lab


def check_digest(t, dg, rtol, atol=0.0, what=""):
    """Compares tensor `t` with a stored digest {shape,sum,abssum,l2,idx,vals}.
    Sampled elements: |a-b| <= atol + rtol*|b|;  l2 / abssum: relative rtol;  sum: rtol*abssum."""
    a = t.detach().cpu().double().reshape(-1).numpy()
    assert list(t.shape) == dg["shape"], f"{what}: shape {list(t.shape)} != {dg['shape']}"
    got = a[dg["idx"]]
    exp = np.array(dg["vals"])
    scale = dg["l2"] / max(np.sqrt(a.size), 1.0)          # rms magnitude of the tensor
    tol = atol + rtol * np.maximum(np.abs(exp), scale)
    bad = np.abs(got - exp) > tol
    assert not bad.any(), f"{what}: sampled elements differ: got {got[bad]} exp {exp[bad]} tol {tol[bad]}"
    l2 = float(np.sqrt((a * a).sum()))
    assert abs(l2 - dg["l2"]) <= rtol * max(dg["l2"], atol) + atol, f"{what}: l2 {l2} vs {dg['l2']}"
    assert abs(np.abs(a).sum() - dg["abssum"]) <= rtol * dg["abssum"] + atol, f"{what}: abssum"
    assert abs(a.sum() - dg["sum"]) <= rtol * dg["abssum"] + atol * a.size, f"{what}: sum {a.sum()} vs {dg['sum']}"


def check_digest_dict(named, dgs, rtol, atol=0.0, what="", skip=()):
    for k, dg in dgs.items():
        if any(s in k for s in skip):
            continue
        assert k in named, f"{what}: missing {k}"
        check_digest(named[k], dg, rtol, atol, f"{what}:{k}")


def rel(a, b):
    return abs(a - b) / max(abs(b), 1e-12)
