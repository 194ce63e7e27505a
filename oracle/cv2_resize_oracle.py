"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the single-person pre-path's ``cv2.resize`` for 8-bit 3-channel frames.

**PARITY UNPINNED.**  Reference call sites: ``SimpleHRNet.py:213-222`` (one frame) and ``:355-366`` (a stack): with
``multiperson=False`` every frame goes through ``cv2.resize(image, (W, H), interpolation=self.interpolation)`` (default
``cv2.INTER_CUBIC``, ``:27``), ``cv2.cvtColor(BGR2RGB)`` and ``ToTensor`` + ``Normalize``.  The arithmetic lives in a
third-party dependency that is neither vendored in the reference nor installed in this image (``requirements.txt:5``:
``opencv-python>=3.4``, no upper pin), so there is nothing here to generate golden vectors from.  What follows restates the
published generic (non-IPP, non-OpenCL) path of ``modules/imgproc/src/resize.cpp`` for ``CV_8UC3``:

* sample positions ``fx = (float)((dx + 0.5) * scale_x - 0.5)``, ``sx = cvFloor(fx)``, ``fx -= sx`` with
  ``scale_x = 1 / ((double)W / src_w)`` (``resize()``: the ``xofs`` / ``alpha`` loop);
* ``INTER_CUBIC``: ``interpolateCubic`` (A = -0.75, float32), coefficients ``saturate_cast<short>(c * 2048)`` (round half to
  even), horizontal pass in int32 with replicate border per tap (``HResizeCubic``), vertical pass and
  ``FixedPtCast<int, uchar, 22>``: ``saturate_cast<uchar>((v + (1 << 21)) >> 22)`` (``VResizeCubic``);
* ``INTER_LINEAR``: coefficients ``(1 - fx, fx) * 2048``; ``sx < 0 -> (0, fx = 0)``, ``sx >= src_w - 1 -> (src_w - 1, fx = 0)``;
  rows clipped; ``VResizeLinear<uchar, int, short, ...>``: ``(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2``;
* ``INTER_NEAREST``: ``sx = min(cvFloor(dx * scale_x), src_w - 1)`` (``resizeNN``).

OpenCV builds that route 8-bit resizes through IPP or OpenCL are documented NOT to be bit-identical to this path; the GPU kernel
(``simple-hrnet_amd/csrc/prepath.hip: resize_frames_kernel``) is pinned to THIS restatement bit for bit
(``tests/test_resize.py``), and this restatement to hand-derivable properties and to the textbook float formula (Keys kernel, a = -0.75,
half-pixel centres, replicate border: never more than one grey level away) only.  Only tests may import this module.
"""
import numpy as np

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2   # = cv2.INTER_*
COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS
MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)  # SimpleHRNet.py:171
STD = np.array([0.229, 0.224, 0.225], dtype=np.float32)


def _sat_short(v32: np.ndarray) -> np.ndarray:
    """saturate_cast<short>(float): cvRound (nearest, ties to even), then clamp"""
    return np.clip(np.rint(v32.astype(np.float64)), -32768, 32767).astype(np.int64)


def _positions(dst: int, src: int):
    scale = 1.0 / (float(dst) / float(src))                                  # double
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    return s, (f - s.astype(np.float32)).astype(np.float32), scale


def _cubic_coeffs(x: np.ndarray) -> np.ndarray:
    """interpolateCubic, float32 step by step (no fused multiply-add)"""
    f = np.float32
    A = f(-0.75)
    x1 = (x + f(1)).astype(np.float32)
    c0 = (((A * x1 - f(5) * A).astype(np.float32) * x1 + f(8) * A).astype(np.float32) * x1 - f(4) * A).astype(np.float32)
    c1 = ((((A + f(2)) * x - (A + f(3))).astype(np.float32) * x).astype(np.float32) * x + f(1)).astype(np.float32)
    xm = (f(1) - x).astype(np.float32)
    c2 = ((((A + f(2)) * xm - (A + f(3))).astype(np.float32) * xm).astype(np.float32) * xm + f(1)).astype(np.float32)
    c3 = (((f(1) - c0).astype(np.float32) - c1).astype(np.float32) - c2).astype(np.float32)
    return np.stack([c0, c1, c2, c3], axis=1)


def taps(dst: int, src: int, interpolation: int):
    """per output position: index of the first tap and the fixed-point coefficients (K = 4 cubic, 2 linear, 1 nearest);
    ``clamp_x`` = the x-side rule of INTER_LINEAR (position pulled inside, fraction dropped)"""
    s, f, scale = _positions(dst, src)
    if interpolation == INTER_NEAREST:
        s = np.minimum(np.floor(np.arange(dst, dtype=np.float64) * scale).astype(np.int64), src - 1)
        return s, np.full((dst, 1), COEF_SCALE, np.int64)
    if interpolation == INTER_CUBIC:
        return s - 1, _sat_short((_cubic_coeffs(f) * np.float32(COEF_SCALE)).astype(np.float32))
    if interpolation == INTER_LINEAR:
        return s, _sat_short((np.stack([np.float32(1) - f, f], axis=1).astype(np.float32) * np.float32(COEF_SCALE)).astype(np.float32))
    raise ValueError("interpolation must be INTER_NEAREST (0), INTER_LINEAR (1) or INTER_CUBIC (2)")


def resize_u8(img: np.ndarray, out_hw, interpolation: int = INTER_CUBIC) -> np.ndarray:
    """``cv2.resize(img, (W, H), interpolation=...)`` for an (h, w, c) uint8 array"""
    img = np.asarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3
    h, w = img.shape[:2]
    H, W = out_hw
    if (H, W) == (h, w):
        return img.copy()
    src = img.astype(np.int64)
    if interpolation == INTER_NEAREST:
        sx, _ = taps(W, w, interpolation)
        sy, _ = taps(H, h, interpolation)
        return img[sy][:, sx]
    x0, ax = taps(W, w, interpolation)
    y0, ay = taps(H, h, interpolation)
    if interpolation == INTER_LINEAR:
        # x side: positions left of the first / right of the last pixel use that pixel alone (resize(): fx = 0, sx = 0 / w - 1)
        lo, hi = x0 < 0, x0 >= w - 1
        x0 = np.where(lo, 0, np.where(hi, w - 1, x0))
        ax = np.where((lo | hi)[:, None], np.array([COEF_SCALE, 0], np.int64)[None, :], ax)
    K = ax.shape[1]
    xi = np.clip(x0[:, None] + np.arange(K)[None, :], 0, w - 1)                    # replicate border, per tap
    yi = np.clip(y0[:, None] + np.arange(K)[None, :], 0, h - 1)
    hor = np.einsum("hwkc,wk->hwc", src[:, xi, :], ax)                              # (h, W, c) int
    rows = hor[yi]                                                                  # (H, K, W, c)
    if interpolation == INTER_CUBIC:
        v = np.einsum("hkwc,hk->hwc", rows, ay)
        out = (v + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS)
    else:
        b0, b1 = ay[:, 0][:, None, None], ay[:, 1][:, None, None]
        out = (((b0 * (rows[:, 0] >> 4)) >> 16) + ((b1 * (rows[:, 1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def single_person_transform(frames_bgr: np.ndarray, out_hw, interpolation: int = INTER_CUBIC) -> np.ndarray:
    """``SimpleHRNet.py:213-222`` / ``:355-366``: resize, BGR -> RGB, ToTensor, Normalize; (n, h, w, 3) uint8 -> (n, 3, H, W) float32"""
    frames_bgr = np.asarray(frames_bgr)
    if frames_bgr.ndim == 3:
        frames_bgr = frames_bgr[None]
    out = np.empty((len(frames_bgr), 3) + tuple(out_hw), np.float32)
    for i, f in enumerate(frames_bgr):
        rgb = resize_u8(f, out_hw, interpolation)[..., ::-1]
        x = rgb.transpose(2, 0, 1).astype(np.float32) / np.float32(255)
        out[i] = (x - MEAN[:, None, None]) / STD[:, None, None]
    return out
