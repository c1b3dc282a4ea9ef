"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's evaluation-time image transform
(data/augmentations.py:28-44 get_valid_transforms / get_test_transforms, applied at data/dataset.py:76-81):

    [h / w > 2: PIL image.rotate(90, expand=True)]  ->  A.Resize(height, width)  ->  A.Normalize(mean, std)  ->  ToTensorV2

PARITY UNPINNED: the arithmetic lives in third-party packages that are absent from /root/reference and from this image
(albumentations==0.5.2 -> cv2.resize(..., interpolation=cv2.INTER_LINEAR) and albumentations.augmentations.functional.normalize;
requirements.txt pins opencv-python==4.5.1.48).  What is restated here is their published algorithm:
  * cv2 INTER_LINEAR on uint8 (imgproc/resize.cpp): pixel centres fx = (dx + 0.5) * scale - 0.5, taps clamped at the borders,
    coefficients rounded to 11-bit fixed point (cvRound(c * 2048), half to even), horizontal pass in int32, vertical pass
    (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2; an exact 2x2 downscale is routed to the INTER_AREA fast path
    ((a + b + c + d + 2) >> 2);
  * Normalize: (img - mean * 255) * (1 / (std * 255)) in float32; ToTensorV2: HWC -> CHW.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import numpy as np

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
COEF_BITS = 11
ONE = 1 << COEF_BITS


def _taps(dst, src):
    """per destination index: (s0, s1, a0, a1) with the fixed-point weights of cv2's linear interpolation"""
    scale = float(src) / float(dst)
    s0 = np.zeros(dst, np.int64); s1 = np.zeros(dst, np.int64)
    a0 = np.zeros(dst, np.int64); a1 = np.zeros(dst, np.int64)
    for d in range(dst):
        f = np.float32((d + 0.5) * scale - 0.5)
        s = int(np.floor(f))
        f = np.float32(f - s)
        if s < 0:
            s, f = 0, np.float32(0)
        if s >= src - 1:
            s, f = src - 1, np.float32(0)
        c0, c1 = np.float32(1.0) - f, f
        # saturate_cast<short>(float) == cvRound: round half to even
        a0[d] = int(np.rint(np.float32(c0 * np.float32(ONE))))
        a1[d] = int(np.rint(np.float32(c1 * np.float32(ONE))))
        s0[d], s1[d] = s, min(s + 1, src - 1)
    return s0, s1, a0, a1


def resize_linear_u8(img, height, width):
    """img uint8 [h, w] or [h, w, c] -> uint8 [height, width(, c)]"""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w, c = img.shape
    if (h, w) == (height, width):
        out = img.copy()
    elif h == 2 * height and w == 2 * width:  # INTER_AREA fast path
        s = img.astype(np.int64)
        out = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    else:
        x0, x1, ax0, ax1 = _taps(width, w)
        y0, y1, ay0, ay1 = _taps(height, h)
        s = img.astype(np.int64)
        hor = s[:, x0, :] * ax0[None, :, None] + s[:, x1, :] * ax1[None, :, None]          # [h, width, c] int
        top, bot = hor[y0], hor[y1]
        v = (((ay0[:, None, None] * (top >> 4)) >> 16) + ((ay1[:, None, None] * (bot >> 4)) >> 16) + 2) >> 2
        out = np.clip(v, 0, 255).astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def rotate90_if_tall(img):
    """data/dataset.py:77-79: PIL rotate(90, expand=True) (counter-clockwise) when h / w > 2"""
    h, w = img.shape[:2]
    return np.rot90(img, 1).copy() if h / w > 2 else img


def normalize_to_chw(img_u8, channels):
    """A.Normalize(mean, std, max_pixel_value=255) + ToTensorV2 -> float32 [C, H, W].  A one-channel image uses the first
    mean / std entry (albumentations itself only accepts the 3-channel case: the reference trains with data.rgb = 3)."""
    x = np.asarray(img_u8).astype(np.float32)
    if x.ndim == 2:
        x = x[:, :, None]
    mean = np.array(MEAN[:channels], np.float32) * np.float32(255.0)
    den = np.reciprocal(np.array(STD[:channels], np.float32) * np.float32(255.0))
    x = (x - mean) * den
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def preprocess(img_u8, height, width):
    """one image uint8 [h, w(, c)] -> float32 [C, height, width] (the evaluation transform)"""
    img = rotate90_if_tall(np.asarray(img_u8))
    c = 1 if img.ndim == 2 else img.shape[2]
    return normalize_to_chw(resize_linear_u8(img, height, width), c)
