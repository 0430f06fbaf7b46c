"""Input transform: the oracle restatement and the host-side implementation against what Pillow -
the library behind the reference's transforms.Resize(64) - returns (tests/golden/resize_u8.json)."""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_golden_resize import inputs  # noqa: E402  (input generator only; needs numpy, not Pillow)


def _gold():
    return json.load(open(os.path.join(HERE, "golden", "resize_u8.json")))


def _check(y, case):
    assert hashlib.sha256(np.ascontiguousarray(y).tobytes()).hexdigest() == case["sha256"]
    assert int(y.astype(np.int64).sum()) == case["sum"]
    if "full" in case:
        assert np.array_equal(y, np.asarray(case["full"], dtype=np.uint8))
    for n, c, yy, xx, v in case.get("samples", []):
        assert int(y[n, c, yy, xx]) == v


def test_oracle_resize_matches_pillow():
    from oracle.preprocess_oracle import resize2x_u8
    for case in _gold()["cases"].values():
        _check(resize2x_u8(inputs(case["seed"], case["n"], case["h"])), case)


def test_host_resize_matches_pillow_and_oracle_transform():
    from oracle.preprocess_oracle import transform
    from preprocess.dcgan_data_preprocessor import resize2x_pil_u8
    for case in _gold()["cases"].values():
        x = inputs(case["seed"], case["n"], case["h"])
        y = resize2x_pil_u8(torch.from_numpy(x)).numpy()
        _check(y, case)
    x = inputs(7, 6, 32)
    t = (resize2x_pil_u8(torch.from_numpy(x)).float() / 255.0 - 0.5) / 0.5
    assert np.array_equal(t.numpy(), transform(x))                    # same fp32 operation order as ToTensor + Normalize
    assert float(t.min()) >= -1.0 and float(t.max()) <= 1.0


def test_pillow_still_agrees_when_installed():
    """Where Pillow is importable the fixture is re-derived live (guards against a stale fixture)."""
    try:
        from make_golden_resize import pil_resize
        import PIL  # noqa: F401
    except Exception:
        import pytest
        pytest.skip("Pillow not installed")
    for case in _gold()["cases"].values():
        x = inputs(case["seed"], case["n"], case["h"])
        _check(np.stack([pil_resize(img, 2 * case["h"]) for img in x]), case)
