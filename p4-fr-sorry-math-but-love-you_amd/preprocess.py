"""Evaluation-time image transform on the GPU (image half): the host pipeline of data/dataset.py:76-81 and
data/augmentations.py:28-44 -- rotate tall images, A.Resize (cv2 INTER_LINEAR), A.Normalize, ToTensorV2 -- for a whole batch
of decoded uint8 images of different sizes in ONE kernel launch after ONE host-to-device copy."""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import SatrnError, check

MEAN = (0.485, 0.456, 0.406)   # data/augmentations.py:20,31,40
STD = (0.229, 0.224, 0.225)


class _Desc(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("h", ctypes.c_int32), ("w", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32)]


def preprocess_images(images, height, width, device="cuda", mean=MEAN, std=STD):
    """images: sequence of uint8 arrays / tensors, each [h, w] (grayscale, data.rgb = 1) or [h, w, 3] (RGB), any sizes
    -> float32 [B, C, height, width] on `device`, normalised, ready for model(input, ...)."""
    lib = _lib.load()
    device = torch.device(device)
    if device.type != "cuda":
        raise SatrnError("preprocess_images runs on the GPU (no CPU fallback)")
    arrs = []
    for im in images:
        a = im.detach().cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
        if a.dtype != np.uint8 or a.ndim not in (2, 3):
            raise SatrnError("preprocess_images takes uint8 [h, w] or [h, w, c] images")
        arrs.append(np.ascontiguousarray(a))
    if not arrs:
        raise SatrnError("preprocess_images: empty batch")
    C = 1 if arrs[0].ndim == 2 else arrs[0].shape[2]
    if C not in (1, 3) or any((1 if a.ndim == 2 else a.shape[2]) != C for a in arrs):
        raise SatrnError("preprocess_images: every image must have the same channel count (1 or 3)")
    # one staging buffer, one upload (each image starts on a 16-byte boundary)
    offs, total = [], 0
    for a in arrs:
        offs.append(total)
        total += (a.size + 15) // 16 * 16
    host = torch.empty(total, dtype=torch.uint8).pin_memory()
    hv = host.numpy()
    for a, o in zip(arrs, offs):
        hv[o:o + a.size] = a.reshape(-1)
    dev = host.to(device, non_blocking=True)
    descs = (_Desc * len(arrs))()
    for i, (a, o) in enumerate(zip(arrs, offs)):
        descs[i] = _Desc(dev.data_ptr() + o, a.shape[0], a.shape[1], a.shape[1] * C, 0)
    dtab = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
    out = torch.empty(len(arrs), C, int(height), int(width), dtype=torch.float32, device=device)
    m3 = (ctypes.c_float * 3)(*(list(mean) + [0, 0, 0])[:3])
    s3 = (ctypes.c_float * 3)(*(list(std) + [1, 1, 1])[:3])
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.satrn_image_preprocess(ctypes.c_void_p(dtab.data_ptr()), len(arrs), C, int(height), int(width), ctypes.c_void_p(out.data_ptr()),
                                     m3, s3, st), "satrn_image_preprocess")
    out._satrn_keep = (dev, dtab)   # the launch reads them asynchronously
    return out
