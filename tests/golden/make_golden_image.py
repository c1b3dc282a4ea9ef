"""Generates tests/golden/image.npz: inputs and expected outputs of the evaluation-time image transform.
The reference implements this step with albumentations / OpenCV (data/augmentations.py:28-44), neither of which is
installed here and neither of which lives under /root/reference, so these vectors come from the numpy restatement of the
published algorithm in oracle/image_oracle.py (PARITY UNPINNED for this row: they fix the restatement, not the reference)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import image_oracle as IO  # noqa: E402


def cases():
    rng = np.random.RandomState(7)
    # (source h, w, channels, target H, W): identity, exact 2x, up- and down-scaling, tall (rotated) inputs, odd sizes
    shapes = [(37, 211, 1, 32, 96), (32, 96, 1, 32, 96), (64, 192, 1, 32, 96), (61, 500, 3, 32, 96), (300, 90, 1, 32, 96),
              (200, 64, 3, 32, 96), (16, 16, 1, 32, 96), (129, 385, 1, 128, 384)]
    out = []
    for h, w, c, H, W in shapes:
        # smooth strokes + noise: something like a scanned formula, full 0..255 range
        yy, xx = np.mgrid[0:h, 0:w]
        base = 127.5 + 127.5 * np.sin(xx * 0.11 + yy * 0.07) * np.cos(yy * 0.05)
        img = np.clip(base[..., None] + rng.randint(-40, 41, size=(h, w, c)), 0, 255).astype(np.uint8)
        out.append((img[:, :, 0] if c == 1 else img, H, W))
    return out


def main():
    d = {}
    for i, (img, H, W) in enumerate(cases()):
        d[f"in{i}"] = img
        d[f"hw{i}"] = np.array([H, W])
        d[f"out{i}"] = IO.preprocess(img, H, W)
        d[f"u8_{i}"] = IO.resize_linear_u8(IO.rotate90_if_tall(img), H, W)
    d["n"] = np.array(len(cases()))
    # a hand-checkable known answer: 2x4 -> 1x2 is the exact-2x path, (a+b+c+d+2)>>2
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "image.npz"), **d)
    print("wrote image.npz", {k: v.shape for k, v in d.items() if k.startswith("out")})


if __name__ == "__main__":
    main()
