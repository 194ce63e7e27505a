"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's crop pre-path.

Reference: ``SimpleHRNet.py:236-278`` (single image, multi-person): per detection -- round the box, correct its
aspect ratio by PADDING (not enlarging), slice the BGR frame as RGB, zero-pad, then
``transforms.Compose([ToPILImage(), Resize((H, W)), ToTensor(), Normalize(mean, std)])`` (``SimpleHRNet.py:167-172``).

torchvision is absent from this image; its four transforms are thin wrappers restated here from their documented
semantics (``ToPILImage``: HWC uint8 ndarray -> RGB image; ``Resize``: ``Image.resize((W, H), BILINEAR)``;
``ToTensor``: CHW ``float32(v) / 255``; ``Normalize``: ``(x - mean) / std`` in float32).  The arithmetic that
matters lives in Pillow, which IS installed (12.2.0), so the restatement of its resampler below
(``pil_bilinear_u8``; libImaging/Resample.c: ``precompute_coeffs``, ``normalize_coeffs_8bpc``,
``ImagingResampleHorizontal_8bpc`` / ``Vertical_8bpc``) is pinned bit-for-bit against ``PIL.Image.resize`` in
``tests/test_prepath.py``.  Only tests, ``__graft_entry__.smoke()`` and the CPU leg of measurements may import this.
"""
import math

import numpy as np

MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)  # SimpleHRNet.py:171
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)
PRECISION_BITS = 32 - 8 - 2  # Resample.c: coefficients of the 8-bit path are 22-bit fixed point


def py_round(x: float) -> int:
    """Python 3 ``round(float)``: nearest integer, ties to even (what ``int(round(x1.item()))`` does, :237-240)."""
    return int(round(float(x)))


def crop_box(det, height: int, width: int):
    """``SimpleHRNet.py:237-272``: (x1, y1, x2, y2) floats -> rounded slice box, reported (padded) box, pad amounts.

    Returns ``(x1, y1, x2, y2), (x1n, y1n, x2n, y2n), (pad_top, pad_bottom, pad_left, pad_right)``."""
    x1, y1, x2, y2 = (py_round(v) for v in det[:4])
    correction_factor = height / width * (x2 - x1) / (y2 - y1)
    pt = pb = pl = pr = 0
    if correction_factor > 1:  # increase y side
        center = y1 + (y2 - y1) // 2
        length = int(round((y2 - y1) * correction_factor))
        x1n, x2n = x1, x2
        y1n, y2n = int(center - length // 2), int(center + length // 2)
        pt, pb = int(abs(y1n - y1)), int(abs(y2n - y2))
    elif correction_factor < 1:
        center = x1 + (x2 - x1) // 2
        length = int(round((x2 - x1) * 1 / correction_factor))
        x1n, x2n = int(center - length // 2), int(center + length // 2)
        y1n, y2n = y1, y2
        pl, pr = abs(x1n - x1), int(abs(x2n - x2))
    else:
        x1n, x2n, y1n, y2n = x1, x2, y1, y2
    return (x1, y1, x2, y2), (x1n, y1n, x2n, y2n), (pt, pb, pl, pr)


def crop_box_clamped(det, height: int, width: int, frame_h: int, frame_w: int):
    """``SimpleHRNet.py:386-407`` (the batch path): the aspect ratio is corrected by ENLARGING the box, clamped to the
    frame; no padding.  Returns the box that is both sliced and reported."""
    x1, y1, x2, y2 = (py_round(v) for v in det[:4])
    correction_factor = height / width * (x2 - x1) / (y2 - y1)
    if correction_factor > 1:
        center = y1 + (y2 - y1) // 2
        length = int(round((y2 - y1) * correction_factor))
        y1 = max(0, center - length // 2)
        y2 = min(frame_h, center + length // 2)
    elif correction_factor < 1:
        center = x1 + (x2 - x1) // 2
        length = int(round((x2 - x1) * 1 / correction_factor))
        x1 = max(0, center - length // 2)
        x2 = min(frame_w, center + length // 2)
    return x1, y1, x2, y2


def _bilinear(x: float) -> float:  # Resample.c: bilinear_filter, support 1.0
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c ``precompute_coeffs`` (box = the whole axis) + ``normalize_coeffs_8bpc``.

    Returns ``bounds[out_size][2]`` (first tap, tap count) and ``kk[out_size][ksize]`` int32 fixed-point weights."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """one 8-bit pass along axis 0 of an (n, m, c) uint8 array -> (out_size, m, c) uint8 (rounded, clipped)."""
    bounds, kk = precompute_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for k in range(n):
            acc += src[xmin + k] * int(kk[xx, k])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """``Image.fromarray(img).resize((out_w, out_h), BILINEAR)`` for HWC uint8: horizontal pass (skipped when the
    width is unchanged), then vertical pass, 8-bit rounding between them (Resample.c ``ImagingResampleInner``)."""
    h, w = img.shape[:2]
    tmp = img
    if w != out_w:
        tmp = _resample_axis0(np.ascontiguousarray(img.transpose(1, 0, 2)), out_w).transpose(1, 0, 2)
    if h != out_h:
        tmp = _resample_axis0(np.ascontiguousarray(tmp), out_h)
    return np.ascontiguousarray(tmp)


def to_tensor_normalize(rgb_u8: np.ndarray) -> np.ndarray:
    """``ToTensor`` then ``Normalize`` (float32 throughout): HWC uint8 -> CHW float32."""
    x = rgb_u8.astype(np.float32) / np.float32(255)
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x.transpose(2, 0, 1)).astype(np.float32)


def prepath(frame_bgr: np.ndarray, dets: np.ndarray, height: int, width: int, resize=pil_bilinear_u8):
    """``SimpleHRNet.py:236-278``: frame (Hf, Wf, 3) uint8 BGR, dets (P, >=4) float -> images (P,3,H,W) float32,
    boxes (P,4) int32 (the padded coordinates the decode scales by)."""
    p = len(dets)
    images = np.empty((p, 3, height, width), np.float32)
    boxes = np.empty((p, 4), np.int32)
    for i, det in enumerate(dets):
        (x1, y1, x2, y2), new, (pt, pb, pl, pr) = crop_box(det, height, width)
        crop = frame_bgr[y1:y2, x1:x2, ::-1]
        if pt or pb or pl or pr:
            crop = np.pad(crop, ((pt, pb), (pl, pr), (0, 0)))
        images[i] = to_tensor_normalize(resize(np.ascontiguousarray(crop), height, width))
        boxes[i] = new
    return images, boxes


def prepath_clamped(frame_bgr: np.ndarray, dets: np.ndarray, height: int, width: int, resize=pil_bilinear_u8):
    """``SimpleHRNet.py:383-412`` for ONE image of the stack: enlarge-and-clamp boxes, slice as RGB, transform."""
    p = len(dets)
    images = np.empty((p, 3, height, width), np.float32)
    boxes = np.empty((p, 4), np.int32)
    for i, det in enumerate(dets):
        x1, y1, x2, y2 = crop_box_clamped(det, height, width, frame_bgr.shape[0], frame_bgr.shape[1])
        images[i] = to_tensor_normalize(resize(np.ascontiguousarray(frame_bgr[y1:y2, x1:x2, ::-1]), height, width))
        boxes[i] = (x1, y1, x2, y2)
    return images, boxes


def pil_resize(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """The real thing (needs Pillow): what torchvision's ``Resize`` calls for a PIL image."""
    from PIL import Image

    return np.asarray(Image.fromarray(img).resize((out_w, out_h), Image.BILINEAR))
