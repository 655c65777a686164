"""Oracle: the four OpenCV calls on the hot path, restated from their published
semantics (opencv-python-headless 4.13.0.90, pinned in the reference's uv.lock;
the module is NOT installed here and its source is not under /root/reference).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED at bit level:
these restate OpenCV's documented/observable behaviour and are pinned only
end-to-end by the reference's tolerance tests (tests/test_beads.py,
tests/test_chip.py scenarios).  Each function is isolated so that a mismatch,
if ever observed against a real OpenCV, is corrected in one place.

Call sites in the reference: utils.py:115 (GaussianBlur), :118-119 (Scharr),
:128-134 (Canny with user gradients, L2gradient=True), :38 (circle, filled).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

TG22 = 13573  # round(tan(22.5 deg) * 2**15), OpenCV's CANNY_SHIFT = 15 fixed-point tangent


def reflect101(idx: np.ndarray, n: int) -> np.ndarray:
    """cv::borderInterpolate(..., BORDER_REFLECT_101): ...cb|abcdefgh|gf..."""
    idx = np.asarray(idx, dtype=np.int64).copy()
    if n == 1:
        return np.zeros_like(idx)
    while True:
        neg = idx < 0
        big = idx >= n
        if not (neg.any() or big.any()):
            return idx
        idx[neg] = -idx[neg]
        idx[big] = 2 * n - 2 - idx[big]


def _pad101(img: np.ndarray, k: int) -> np.ndarray:
    h, w = img.shape
    rr = reflect101(np.arange(-k, h + k), h)
    cc = reflect101(np.arange(-k, w + k), w)
    return img[np.ix_(rr, cc)]


def gaussian_blur5(img: np.ndarray) -> np.ndarray:
    """cv.GaussianBlur(u8, (5,5), 0): sigma<=0 with ksize 5 selects the fixed
    kernel [1,4,6,4,1]/16; for CV_8U OpenCV runs its bit-exact fixed-point path
    (8.8 then 16.16 fixed point), which is exact until the single final rounding
    ``(sum_2d + 128) >> 8`` with sum_2d the [1,4,6,4,1]x[1,4,6,4,1] weighted sum.
    Border: BORDER_REFLECT_101."""
    assert img.dtype == np.uint8 and img.ndim == 2
    if img.size == 0:
        return img.copy()
    p = _pad101(img, 2).astype(np.int32)
    h, w = img.shape
    k = (1, 4, 6, 4, 1)
    horiz = sum(k[j] * p[:, j : j + w] for j in range(5))
    vert = sum(k[i] * horiz[i : i + h, :] for i in range(5))
    return ((vert + 128) >> 8).astype(np.uint8)


def scharr(img: np.ndarray):
    """cv.Scharr(u8, CV_32F, 1, 0) and (0, 1): unscaled 3x3 kernels
    [-3 0 3; -10 0 10; -3 0 3] and its transpose (correlation), REFLECT_101.
    Values are exact integers (|.| <= 4080) held in float32."""
    assert img.dtype == np.uint8 and img.ndim == 2
    p = _pad101(img, 1).astype(np.int32)
    h, w = img.shape
    diff_x = p[:, 2:] - p[:, :-2]  # right - left, all padded rows
    dx = 3 * diff_x[0:h] + 10 * diff_x[1 : h + 1] + 3 * diff_x[2 : h + 2]
    diff_y = p[2:, :] - p[:-2, :]  # below - above, all padded cols
    dy = 3 * diff_y[:, 0:w] + 10 * diff_y[:, 1 : w + 1] + 3 * diff_y[:, 2 : w + 2]
    return dx.astype(np.float32), dy.astype(np.float32)


def canny_thresholds(low_thresh: float, high_thresh: float):
    """Threshold preparation of cv::Canny(dx, dy, ..., L2gradient=true): swap if
    out of order, clamp to 32767, square when positive, cvFloor."""
    lo, hi = float(low_thresh), float(high_thresh)
    if lo > hi:
        lo, hi = hi, lo
    lo, hi = min(32767.0, lo), min(32767.0, hi)
    if lo > 0:
        lo *= lo
    if hi > 0:
        hi *= hi
    return int(np.floor(lo)), int(np.floor(hi))


def canny_nms(dx: np.ndarray, dy: np.ndarray, low: int, high: int) -> np.ndarray:
    """Non-maximum suppression + double threshold of cv::Canny (L2 magnitude).

    Returns the OpenCV map values: 1 = not an edge, 0 = weak candidate,
    2 = strong edge.  Magnitude is the integer dx^2+dy^2, zero outside the
    image.  Direction sectors use the Q15 tangent test with OpenCV's asymmetric
    comparisons (> towards the previous pixel, >= towards the next; strict > on
    both diagonal neighbours)."""
    xs = dx.astype(np.int64)
    ys = dy.astype(np.int64)
    h, w = xs.shape
    mag = xs * xs + ys * ys
    mp = np.zeros((h + 2, w + 2), dtype=np.int64)
    mp[1:-1, 1:-1] = mag

    def nb(di, dj):
        return mp[1 + di : 1 + di + h, 1 + dj : 1 + dj + w]

    ax = np.abs(xs)
    ay = np.abs(ys) << 15
    tg22x = ax * TG22
    tg67x = tg22x + (ax << 16)
    horizontal = ay < tg22x
    vertical = ~horizontal & (ay > tg67x)
    diagonal = ~horizontal & ~vertical
    s_neg = (xs ^ ys) < 0  # s = -1 where signs differ
    keep_h = (mag > nb(0, -1)) & (mag >= nb(0, 1))
    keep_v = (mag > nb(-1, 0)) & (mag >= nb(1, 0))
    # s = +1: compare with (row-1, col-1) and (row+1, col+1); s = -1: (row-1, col+1), (row+1, col-1)
    keep_d = np.where(s_neg, (mag > nb(-1, 1)) & (mag > nb(1, -1)), (mag > nb(-1, -1)) & (mag > nb(1, 1)))
    local_max = (horizontal & keep_h) | (vertical & keep_v) | (diagonal & keep_d)
    cand = (mag > low) & local_max
    out = np.ones((h, w), dtype=np.uint8)
    out[cand] = 0
    out[cand & (mag > high)] = 2
    return out


def canny_hysteresis(nms_map: np.ndarray) -> np.ndarray:
    """8-connected hysteresis: weak candidates (0) connected to a strong pixel (2)
    become edges.  Returns a {0, 1} uint8 map (the reference sets 255 -> 1,
    utils.py:142)."""
    cand = nms_map != 1
    lab, n = ndimage.label(cand, structure=np.ones((3, 3), dtype=bool))
    if n == 0:
        return np.zeros(nms_map.shape, dtype=np.uint8)
    has_strong = np.zeros(n + 1, dtype=bool)
    has_strong[np.unique(lab[nms_map == 2])] = True
    has_strong[0] = False
    return has_strong[lab].astype(np.uint8)


def canny(dx: np.ndarray, dy: np.ndarray, low_thresh: float, high_thresh: float) -> np.ndarray:
    """cv.Canny(dx.astype(int16), dy.astype(int16), t1, t2, L2gradient=True) != 0."""
    low, high = canny_thresholds(low_thresh, high_thresh)
    return canny_hysteresis(canny_nms(dx.astype(np.int16), dy.astype(np.int16), low, high))


def filled_circle_mask(shape, center_rc, radius: int) -> np.ndarray:
    """cv.circle(img, (cx, cy), radius, 1, thickness=-1) as a bool mask.

    OpenCV's filled circle rasterises the midpoint (Bresenham) circle and fills
    each scanline between its extreme perimeter pixels: the pixel set is
    {(dy, dx): |dx| <= xmax(|dy|)} with xmax taken from the circle walk below
    (cv::Circle in drawing.cpp: err/plus/minus updates)."""
    h, w = shape
    cy, cx = int(center_rc[0]), int(center_rc[1])
    mask = np.zeros((h, w), dtype=bool)
    if radius < 0:
        return mask
    half = np.full(radius + 1, -1, dtype=np.int64)  # half[|dy|] = max |dx|
    err, dx, dy = 0, radius, 0
    plus, minus = 1, (radius << 1) - 1
    while dx >= dy:
        half[dy] = max(half[dy], dx)  # rows cy +- dy get the span [cx-dx, cx+dx]
        half[dx] = max(half[dx], dy)  # rows cy +- dx get the span [cx-dy, cx+dy]
        dy += 1
        err += plus
        plus += 2
        step = -1 if err > 0 else 0  # OpenCV: mask = (err <= 0) - 1
        err -= minus & step
        dx += step
        minus -= step & 2
    for ady in range(radius + 1):
        hw = int(half[ady])
        if hw < 0:
            continue
        for yy in {cy - ady, cy + ady}:
            if 0 <= yy < h:
                x0, x1 = max(cx - hw, 0), min(cx + hw, w - 1)
                if x0 <= x1:
                    mask[yy, x0 : x1 + 1] = True
    return mask
