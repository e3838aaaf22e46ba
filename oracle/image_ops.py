"""Test oracle (NOT product code) for the image side of the data pipeline:
`Resize(scale=(1333, 800), keep_ratio=True)` + `RandomFlip` on uint8 images as the reference's transforms perform them
through ``mmcv.imrescale(..., interpolation='bilinear', backend='cv2')`` / ``mmcv.imflip``.

cv2 is not in this image, so OpenCV's 8-bit bilinear resize is RESTATED from its published algorithm
(modules/imgproc/src/resize.cpp: half-pixel centres, 11-bit fixed-point coefficients, the two-pass
HResizeLinear / VResizeLinear<uchar,int,short> rounding) -- **parity unpinned** against cv2 itself."""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def linear_coeffs(src: int, dst: int):
    """(ofs[dst] int32, coef[dst,2] int16) of cv2's INTER_LINEAR along one axis"""
    scale = 1.0 / (float(dst) / float(src))                 # double, as resize() computes scale_x from inv_scale_x
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = f - s.astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    c1 = np.rint(f.astype(np.float64) * COEF_SCALE).astype(np.int64)       # saturate_cast<short>(float * 2048): round half even
    c0 = np.rint((1.0 - f).astype(np.float32).astype(np.float64) * COEF_SCALE).astype(np.int64)
    return s, np.stack([c0, c1], 1).astype(np.int16)


def resize_linear_u8(img: np.ndarray, new_w: int, new_h: int) -> np.ndarray:
    """img uint8 [H, W, C] -> uint8 [new_h, new_w, C]"""
    H, W, C = img.shape
    xo, xa = linear_coeffs(W, new_w)
    yo, ya = linear_coeffs(H, new_h)
    x1 = np.minimum(xo + 1, W - 1)
    y1 = np.minimum(yo + 1, H - 1)
    src = img.astype(np.int32)
    rows = src[:, xo, :] * xa[:, 0].astype(np.int32)[None, :, None] + src[:, x1, :] * xa[:, 1].astype(np.int32)[None, :, None]
    s0, s1 = rows[yo], rows[y1]
    b0, b1 = ya[:, 0].astype(np.int32)[:, None, None], ya[:, 1].astype(np.int32)[:, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def rescale_size(w: int, h: int, scale=(1333, 800)):
    f = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return int(w * float(f) + 0.5), int(h * float(f) + 0.5)


def resize_flip(img: np.ndarray, scale=(1333, 800), flip: bool = False):
    """-> (uint8 [h', w', C], (w_scale, h_scale)): Resize(keep_ratio) then horizontal RandomFlip"""
    H, W, _ = img.shape
    nw, nh = rescale_size(W, H, scale)
    out = resize_linear_u8(img, nw, nh)
    if flip:
        out = out[:, ::-1]
    return np.ascontiguousarray(out), (nw / W, nh / H)
