"""GPU: satrn_amd.preprocess_images (one launch for a batch of variable-size uint8 images: rotate tall, cv2-style fixed-point
bilinear resize, normalise, CHW) against the CPU oracle -- the resized uint8 image bit-exactly, the float tensor to 1e-6."""
import os

import numpy as np
import pytest
import torch

from oracle import image_oracle as IO

pytestmark = pytest.mark.gpu


def test_preprocess_batch_matches_oracle(golden_dir):
    import satrn_amd
    z = np.load(os.path.join(golden_dir, "image.npz"))
    n = int(z["n"])
    groups = {}
    for i in range(n):
        key = (1 if z[f"in{i}"].ndim == 2 else 3, int(z[f"hw{i}"][0]), int(z[f"hw{i}"][1]))
        groups.setdefault(key, []).append(i)
    for (C, H, W), idx in groups.items():   # one launch per (channels, target size) group of variable-size inputs
        out = satrn_amd.preprocess_images([z[f"in{i}"] for i in idx], H, W).cpu().numpy()
        assert out.shape == (len(idx), C, H, W)
        mean = np.array(IO.MEAN[:C], np.float32) * 255
        std = np.array(IO.STD[:C], np.float32) * 255
        for k, i in enumerate(idx):
            assert np.abs(out[k] - z[f"out{i}"]).max() < 2e-6, i
            # undo the normalisation: the integer image underneath must be the oracle's, bit for bit
            u8 = np.rint(out[k] * std[:, None, None] + mean[:, None, None]).astype(np.int64)
            ref = z[f"u8_{i}"] if z[f"u8_{i}"].ndim == 3 else z[f"u8_{i}"][:, :, None]
            assert np.array_equal(u8, ref.transpose(2, 0, 1)), i


def test_preprocess_random_sizes_and_model_accepts_it():
    import satrn_amd
    rng = np.random.RandomState(3)
    imgs = [rng.randint(0, 256, size=(int(rng.randint(8, 300)), int(rng.randint(8, 700))), dtype=np.uint8) for _ in range(9)]
    imgs.append(rng.randint(0, 256, size=(256, 768), dtype=np.uint8))   # exact 2x
    imgs.append(rng.randint(0, 256, size=(128, 384), dtype=np.uint8))   # identity
    out = satrn_amd.preprocess_images([torch.from_numpy(a) for a in imgs], 128, 384)
    for k, a in enumerate(imgs):
        assert np.abs(out[k].cpu().numpy() - IO.preprocess(a, 128, 384)).max() < 2e-6, (k, a.shape)
    with pytest.raises(satrn_amd.SatrnError):
        satrn_amd.preprocess_images([np.zeros((4, 4), np.float32)], 8, 8)
