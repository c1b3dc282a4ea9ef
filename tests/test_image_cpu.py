"""CPU: the image-transform oracle (oracle/image_oracle.py) against hand-computed known answers and the committed vectors."""
import os

import numpy as np

from oracle import image_oracle as IO


def test_known_answers():
    # identity
    a = np.arange(12, dtype=np.uint8).reshape(3, 4)
    assert (IO.resize_linear_u8(a, 3, 4) == a).all()
    # exact 2x downscale = 2x2 box mean, rounded half up
    b = np.array([[0, 2, 10, 20], [4, 6, 30, 41]], np.uint8)
    assert IO.resize_linear_u8(b, 1, 2).tolist() == [[3, 25]]
    # 2x upscale of a 1x2 row: centres at -0.25, 0.25, 0.75, 1.25 -> clamp, 3/4:1/4, 1/4:3/4, clamp
    c = np.array([[0, 200]], np.uint8)
    assert IO.resize_linear_u8(c, 1, 4).tolist() == [[0, 50, 150, 200]]
    # constant images stay constant under any scale (the fixed-point weights sum to 2048)
    d = np.full((17, 5), 201, np.uint8)
    assert (IO.resize_linear_u8(d, 128, 384) == 201).all()
    # normalisation of mid-grey, one channel
    x = IO.normalize_to_chw(np.full((2, 2), 128, np.uint8), 1)
    assert x.shape == (1, 2, 2) and abs(float(x[0, 0, 0]) - (128 - 0.485 * 255) / (0.229 * 255)) < 1e-6
    # tall image is turned counter-clockwise: the right column becomes the top row
    t = np.arange(10 * 3, dtype=np.uint8).reshape(10, 3)
    r = IO.rotate90_if_tall(t)
    assert r.shape == (3, 10) and (r[0] == t[:, 2]).all()


def test_oracle_reproduces_committed_vectors(golden_dir):
    z = np.load(os.path.join(golden_dir, "image.npz"))
    for i in range(int(z["n"])):
        H, W = (int(v) for v in z[f"hw{i}"])
        out = IO.preprocess(z[f"in{i}"], H, W)
        assert out.shape == z[f"out{i}"].shape and np.array_equal(out, z[f"out{i}"])
