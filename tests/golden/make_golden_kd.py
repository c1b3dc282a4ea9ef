"""Golden vector for the knowledge-distillation loss: the reference's own loss_fn_kd
(train_modules/train_distillation.py:49-55) run on CPU in the authoring container.

    python tests/golden/make_golden_kd.py   ->  tests/golden/kd.npz  (inputs are regenerated from seeds; data only)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G  # noqa: E402
from oracle import satrn_oracle as O  # noqa: E402


def inputs(B, T, V, seed):
    s = O.det_tensor((B, T, V), seed, 4.0)
    t = O.det_tensor((B, T, V), seed + 1, 6.0)
    lab = (O.det_tensor((B, T), seed + 2, 1.0).abs() * 1e4).long() % V
    lab[:, -2:] = O.PAD_ID  # padded tail: the reference's KD cross-entropy does NOT ignore PAD
    return s, t, lab


def main():
    G.import_reference()
    import types
    tv = types.ModuleType("torchvision")  # absent third-party package, not on the arithmetic path (only `transforms` is named)
    tv.transforms = types.ModuleType("torchvision.transforms")
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.transforms", tv.transforms)
    try:
        from train_modules.train_distillation import loss_fn_kd
    except Exception as e:  # the module pulls the whole training stack; fall back to exec of the one function's module
        raise SystemExit(f"cannot import the reference's train_distillation: {e!r}")
    out = {}
    for name, (B, T, V, seed, temp, alpha) in dict(a=(3, 7, 245, 90, 10, 0.1), b=(2, 5, 245, 95, 4, 0.5)).items():
        s, t, lab = inputs(B, T, V, seed)
        s.requires_grad_(True)
        loss = loss_fn_kd(s.transpose(1, 2), lab, t.transpose(1, 2), T=temp, alpha=alpha)
        loss.backward()
        out[name + "_meta"] = np.array([B, T, V, seed, temp], dtype=np.int64)
        out[name + "_alpha"] = np.array(alpha, dtype=np.float64)
        out[name + "_loss"] = np.array(loss.item(), dtype=np.float64)
        out[name + "_grad"] = s.grad.numpy()
        print(name, loss.item(), s.grad.abs().max().item())
    np.savez_compressed(os.path.join(HERE, "kd.npz"), **out)


if __name__ == "__main__":
    main()
