"""ctypes loader for the oracle's C restatement (``oracle/c/ref_port.c``).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``): loaded only by ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg.  Built by
``make -C oracle`` (``__graft_entry__.build()`` runs it) into ``oracle/_build``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libref_port.so")
_LIB = None

_P = C.c_void_p
_SIG = {
    "ref_flatfield_correct_u16": (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_double, C.c_double, _P]),
    "ref_to_uint8_u16": (None, [_P, C.c_int64, _P]),
    "ref_gaussian_blur5": (None, [_P, C.c_int, C.c_int, _P]),
    "ref_scharr": (None, [_P, C.c_int, C.c_int, _P, _P]),
    "ref_quantile_f32": (C.c_float, [_P, C.c_int64, C.c_double]),
    "ref_canny_thresholds": (None, [C.c_double, C.c_double, _P, _P]),
    "ref_canny_nms": (None, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_int64, _P]),
    "ref_canny_hysteresis": (None, [_P, C.c_int, C.c_int, _P]),
    "ref_edge_stage": (None, [_P, C.c_int, C.c_int, C.c_double, C.c_double, _P, _P, _P, _P, _P]),
    "ref_grid_array": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "ref_circle_points": (C.c_int, [C.c_int, C.c_int, _P]),
    "ref_filled_circle_points": (C.c_int, [C.c_int, _P]),
    "ref_circle_labels": (None, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "ref_bounding_box": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "ref_draw_uniform32": (C.c_uint32, [C.c_uint64, C.c_uint64, C.c_int]),
    "ref_circumcircle": (None, [_P, _P, _P, _P]),
    "ref_mean_grad": (None, [_P, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "ref_candidate_circles_from_picks": (None, [_P, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_int64, _P]),
    "ref_filter_neighbors": (None, [_P, C.c_int, C.c_int, _P]),
    "ref_find_circles": (C.c_int64, [_P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.c_int64, C.c_int, C.c_int,
                                     C.c_float, C.c_int, C.c_uint64, _P, _P, C.c_int64]),
    "ref_bead_assay": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                   C.c_int64, C.c_float, _P, C.c_int, C.c_uint64, _P, C.c_int64, _P, _P, _P, _P, _P, _P,
                                   _P]),
    "ref_run_stack": (C.c_int64, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_double, C.c_double, C.c_int, C.c_int,
                                  C.c_int, C.c_double, C.c_double, C.c_int64, C.c_float, _P, C.c_int64, C.c_int, _P,
                                  _P]),
    "ref_max_threads": (C.c_int, []),
}


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(_PATH):
            subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
        _LIB = C.CDLL(_PATH)
        for name, (res, args) in _SIG.items():
            fn = getattr(_LIB, name)
            fn.restype, fn.argtypes = res, args
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(_P)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def flatfield_correct(tiles: np.ndarray, flatfield=1.0, darkfield=0.0) -> np.ndarray:
    t = _c(tiles, np.uint16)
    flat_img = None if np.isscalar(flatfield) else _c(flatfield, np.float32)
    plane = int(flat_img.size) if flat_img is not None else int(t.shape[-1] * t.shape[-2])
    out = np.empty_like(t)
    rc = lib().ref_flatfield_correct_u16(_p(t), t.size, plane, _p(flat_img), float(flatfield) if flat_img is None else 1.0,
                                         float(darkfield), _p(out))
    assert rc == 0
    return out


def to_uint8(img: np.ndarray) -> np.ndarray:
    a = _c(img, np.uint16)
    out = np.empty(a.shape, dtype=np.uint8)
    lib().ref_to_uint8_u16(_p(a), a.size, _p(out))
    return out


def gaussian_blur5(img: np.ndarray) -> np.ndarray:
    a = _c(img, np.uint8)
    out = np.empty_like(a)
    lib().ref_gaussian_blur5(_p(a), a.shape[0], a.shape[1], _p(out))
    return out


def scharr(img: np.ndarray):
    a = _c(img, np.uint8)
    dx = np.empty(a.shape, dtype=np.int16)
    dy = np.empty(a.shape, dtype=np.int16)
    lib().ref_scharr(_p(a), a.shape[0], a.shape[1], _p(dx), _p(dy))
    return dx, dy


def quantile_f32(v: np.ndarray, q: float) -> np.float32:
    a = _c(v, np.float32).ravel()
    return np.float32(lib().ref_quantile_f32(_p(a), a.size, float(q)))


def canny(dx, dy, lo: float, hi: float) -> np.ndarray:
    dx, dy = _c(dx, np.int16), _c(dy, np.int16)
    low, high = C.c_int64(), C.c_int64()
    lib().ref_canny_thresholds(float(lo), float(hi), C.byref(low), C.byref(high))
    h, w = dx.shape
    m = np.empty((h, w), dtype=np.uint8)
    lib().ref_canny_nms(_p(dx), _p(dy), h, w, low.value, high.value, _p(m))
    e = np.empty((h, w), dtype=np.uint8)
    lib().ref_canny_hysteresis(_p(m), h, w, _p(e))
    return e


def edge_stage(img_u8: np.ndarray, low_q: float, high_q: float):
    a = _c(img_u8, np.uint8)
    h, w = a.shape
    blur, edges = np.empty((h, w), np.uint8), np.empty((h, w), np.uint8)
    dx, dy = np.empty((h, w), np.int16), np.empty((h, w), np.int16)
    lohi = np.zeros(2, np.float32)
    lib().ref_edge_stage(_p(a), h, w, float(low_q), float(high_q), _p(blur), _p(dx), _p(dy), _p(edges), _p(lohi))
    return blur, dx, dy, edges, (float(lohi[0]), float(lohi[1]))


def grid_array(edges: np.ndarray, grid: int):
    e = _c(edges, np.uint8)
    h, w = e.shape
    gr, gc = -(-h // grid), -(-w // grid)
    starts, counts = np.empty((gr, gc), np.int64), np.empty((gr, gc), np.int64)
    n = lib().ref_grid_array(_p(e), h, w, grid, None, _p(starts), _p(counts))
    coords = np.empty((n, 2), np.int32)
    lib().ref_grid_array(_p(e), h, w, grid, _p(coords), _p(starts), _p(counts))
    return coords, starts, counts


def circle_points(r: int, four_connected: bool = False) -> np.ndarray:
    buf = np.empty((8 * (r + 1) + 4, 2), np.int32)
    n = lib().ref_circle_points(r, int(four_connected), _p(buf))
    return buf[:n].copy()


def filled_circle_points(r: int) -> np.ndarray:
    buf = np.empty(((2 * r + 1) ** 2, 2), np.int32)
    n = lib().ref_filled_circle_points(r, _p(buf))
    if n < 0:
        raise ValueError("filled_circle_points is undefined for r < 2 in the reference")
    return buf[:n].copy()


def circle_labels(circles: np.ndarray, h: int, w: int) -> np.ndarray:
    c = _c(circles, np.int32).reshape(-1, 3)
    out = np.empty((h, w), np.int32)
    lib().ref_circle_labels(_p(c), len(c), h, w, _p(out))
    return out


def bounding_box(x, y, length, width, height):
    out = (C.c_int * 4)()
    lib().ref_bounding_box(x, y, length, width, height, out)
    return tuple(out)


def circumcircles(p0, p1, p2) -> np.ndarray:
    p0, p1, p2 = (_c(p, np.int32).reshape(-1, 2) for p in (p0, p1, p2))
    out = np.empty((len(p0), 3), np.float32)
    f = lib().ref_circumcircle
    for i in range(len(p0)):
        f(_p(p0[i : i + 1]), _p(p1[i : i + 1]), _p(p2[i : i + 1]), _p(out[i : i + 1]))
    return out


def mean_grad(angles, edges, centers, r: int) -> np.ndarray:
    """Unpadded arrays; centres in image coordinates (out-of-image pixels count as no edge)."""
    a, e = _c(angles, np.float32), _c(edges, np.uint8)
    c = _c(centers, np.int32).reshape(-1, 2)
    out = np.empty(len(c), np.float32)
    lib().ref_mean_grad(_p(a), _p(e), a.shape[0], a.shape[1], _p(c), len(c), r, _p(out))
    return out


def candidate_circles_from_picks(edges, grid, i0, j1, j2) -> np.ndarray:
    e = _c(edges, np.uint8)
    i0, j1, j2 = (_c(v, np.int64).ravel() for v in (i0, j1, j2))
    out = np.empty((len(i0), 3), np.float32)
    if len(i0):
        lib().ref_candidate_circles_from_picks(_p(e), e.shape[0], e.shape[1], grid, _p(i0), _p(j1), _p(j2), len(i0), _p(out))
    return out


def filter_neighbors(circles: np.ndarray, min_dist: int) -> np.ndarray:
    c = _c(circles, np.int32).reshape(-1, 3)
    keep = np.ones(len(c), np.uint8)
    lib().ref_filter_neighbors(_p(c), len(c), min_dist, _p(keep))
    return keep.astype(bool)


def find_circles(img_u8, low_q, high_q, grid, num_iter, min_r, max_r, min_roundness, min_dist, seed=0, cap=1 << 20):
    a = _c(img_u8, np.uint8)
    circles, scores = np.empty((cap, 3), np.int32), np.empty(cap, np.float32)
    n = lib().ref_find_circles(_p(a), a.shape[0], a.shape[1], float(low_q), float(high_q), grid, int(num_iter), min_r,
                               max_r, float(min_roundness), min_dist, seed & 0xFFFFFFFFFFFFFFFF, _p(circles), _p(scores), cap)
    if n < 0:
        raise RuntimeError("circle capacity exceeded")
    return circles[:n].copy(), scores[:n].copy()


def bead_assay(image, min_r, max_r, roi_len, low_q=0.1, high_q=0.9, num_iter=5_000_000, min_roundness=0.3,
               search_channels=(0,), seed=0, cap=1 << 14, want_roi=True):
    """image (C, H, W) uint16 -> dict like oracle.ref_pipeline.find_beads + roi_reduce for T = 1."""
    img = _c(image, np.uint16)
    n_c, h, w = img.shape
    sc = np.asarray(search_channels, np.int32)
    beads = np.empty((cap, 3), np.int32)
    roi = fg = bg = None
    fsum, bsum = np.zeros((cap, n_c), np.int64), np.zeros((cap, n_c), np.int64)
    fcnt, bcnt = np.zeros(cap, np.int64), np.zeros(cap, np.int64)
    if want_roi:  # np.empty: only the pages of the beads actually found are ever touched
        roi = np.empty((cap, n_c, roi_len, roi_len), np.uint16)
        fg = np.empty((cap, roi_len, roi_len), np.uint8)
        bg = np.empty((cap, roi_len, roi_len), np.uint8)
    m = lib().ref_bead_assay(_p(img), n_c, h, w, min_r, max_r, roi_len, float(low_q), float(high_q), int(num_iter),
                             float(min_roundness), _p(sc), len(sc), seed & 0xFFFFFFFFFFFFFFFF, _p(beads), cap, _p(roi),
                             _p(fg), _p(bg), _p(fsum), _p(bsum), _p(fcnt), _p(bcnt))
    if m < 0:
        raise RuntimeError("bead capacity exceeded")
    out = {"beads": beads[:m].copy(), "fg_sum": fsum[:m].copy(), "bg_sum": bsum[:m].copy(), "fg_count": fcnt[:m].copy(),
           "bg_count": bcnt[:m].copy()}
    if roi is not None:
        out["roi"], out["fg"], out["bg"] = roi[:m].copy(), fg[:m].astype(bool), bg[:m].astype(bool)
    return out


def run_stack(stack, flatfield, darkfield, min_r, max_r, roi_len, seeds, low_q=0.1, high_q=0.9, num_iter=5_000_000,
              min_roundness=0.3, cap=1 << 14, n_threads=0):
    """stack (T, C, H, W) uint16, mode P.  Returns (total beads, per-assay counts, per-assay checksums)."""
    s = _c(stack, np.uint16)
    n_t, n_c, h, w = s.shape
    flat_img = None if np.isscalar(flatfield) else _c(flatfield, np.float32)
    sd = np.asarray([x & 0xFFFFFFFFFFFFFFFF for x in seeds], dtype=np.uint64)
    counts, sums = np.zeros(n_t, np.int64), np.zeros(n_t, np.int64)
    total = lib().ref_run_stack(_p(s), n_t, n_c, h, w, _p(flat_img), float(flatfield) if flat_img is None else 1.0,
                                float(darkfield), min_r, max_r, roi_len, float(low_q), float(high_q), int(num_iter),
                                float(min_roundness), _p(sd), cap, n_threads, _p(counts), _p(sums))
    if total < 0:
        raise RuntimeError("bead capacity exceeded")
    return int(total), counts, sums


def max_threads() -> int:
    return int(lib().ref_max_threads())
