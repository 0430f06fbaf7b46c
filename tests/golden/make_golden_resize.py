"""Makes tests/golden/resize_u8.json: what Pillow's bilinear resize (the call behind transforms.Resize(64) on a PIL image,
reference preprocess/dcgan_data_preprocessor.py:39) returns for seeded inputs.  Run where Pillow is installed:
    python tests/golden/make_golden_resize.py
Inputs are regenerated from the seeds by the tests; the fixture holds only digests, a full small case and samples."""
import hashlib
import json
import os

import numpy as np
import PIL
from PIL import Image


def pil_resize(chw, size):
    im = Image.fromarray(np.ascontiguousarray(chw.transpose(1, 2, 0)))
    return np.asarray(im.resize((size, size), Image.BILINEAR)).transpose(2, 0, 1)


def inputs(seed, n, h):
    rs = np.random.RandomState(seed)
    x = rs.randint(0, 256, size=(n, 3, h, h)).astype(np.uint8)
    x[0] = 255                                                        # saturated
    yy, xx = np.mgrid[0:h, 0:h]
    x[1] = ((xx * 255) // (h - 1)).astype(np.uint8)                   # horizontal ramp
    x[2] = (((xx + yy) % 2) * 255).astype(np.uint8)                   # checkerboard (worst case for the rounding)
    return x


def main():
    out = {"pillow": PIL.__version__, "cases": {}}
    for name, seed, n, h in (("cifar_like", 7, 6, 32), ("small", 3, 4, 8)):
        x = inputs(seed, n, h)
        y = np.stack([pil_resize(img, 2 * h) for img in x])
        case = {"seed": seed, "n": n, "h": h, "sha256": hashlib.sha256(y.tobytes()).hexdigest(),
                "sum": int(y.astype(np.int64).sum())}
        if h == 8:
            case["full"] = y.tolist()
        else:
            rs = np.random.RandomState(99)
            pts = [(int(rs.randint(n)), int(rs.randint(3)), int(rs.randint(2 * h)), int(rs.randint(2 * h))) for _ in range(64)]
            pts += [(3, 0, 0, 0), (3, 1, 63, 63), (4, 2, 0, 63), (5, 0, 63, 0), (3, 2, 1, 62)]
            case["samples"] = [[*p, int(y[p])] for p in pts]
        out["cases"][name] = case
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_u8.json")
    json.dump(out, open(path, "w"))
    print("wrote", path)


if __name__ == "__main__":
    main()
