"""ctypes binding of the C-ABI hot-path library (``include/magnify_hip.h``).

The library is built in-tree by ``magnify_amd/csrc/Makefile`` (``__graft_entry__.build()``)
into ``magnify_amd/_lib/libmagnify_hip.so``.  There is NO CPU fallback: if the library is
missing every product entry point raises ``RuntimeError``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_lib", "libmagnify_hip.so")

MG_U8, MG_U16, MG_F32, MG_F64 = 0, 1, 2, 3
MG_NO_EDGE = 100.0
MG_SCORE_SKIPPED = -2.0

_p = C.c_void_p
_i = C.c_int
_l = C.c_int64
_d = C.c_double
_f = C.c_float

# name -> argtypes, exactly the prototypes of include/magnify_hip.h
PROTOTYPES = {
    "mg_version": [],
    "mg_circle_points": [_i, _i, _p, _i],
    "mg_disk_halfwidths": [_i, _p],
    "mg_cv_disk_halfwidths": [_i, _p],
    "mg_perimeter_table": [_i, _i, _p, _p, _p, _i],
    "mg_flatfield_max": [_p, _i, _l, _i, _i, _i, _d, _p, _i, _d, _p, _i, _p, _p, _l, _p],
    "mg_flatfield_max_scratch_floats": [_i, _i, _i],
    "mg_marker_table": [_p, _l, _p, _i, _i, _i, _p, _p, _i, _i, _i, _p, _p],
    "mg_flatfield_bound": [_p, _i, _i, _i, _i, _p, _l, _p],
    "mg_flatfield_is_identity": [_i, _d, _p, _d, _p],
    "mg_flatfield_apply_stitch": [_p, _i, _l, _i, _i, _i, _i, _i, _i, _i, _d, _p, _i, _d, _p, _i, _p, _p, _p, _p],
    "mg_plane_minmax": [_p, _i, _i, _l, _i, _i, _l, _p, _p],
    "mg_to_uint8_blur": [_p, _i, _i, _l, _i, _i, _l, _p, _p, _p, _p],
    "mg_scharr_hist": [_p, _i, _i, _i, _i, _p, _p, _p, _l, _p],
    "mg_scharr_hist_scratch_words": [_i, _i, _i, _i],
    "mg_to_uint8_blur_hist": [_p, _i, _i, _l, _i, _i, _l, _p, _p, _p, _p, _p, _l, _p],
    "mg_blur_hist_scratch_words": [_i, _i, _i],
    "mg_edge_thresholds": [_p, _i, _p, _f, _f, _p, _p, _p, _p, _p, _p],
    "mg_edge_thresholds_window": [_p, _i, _i, _f, _f, _p, _p, _p, _p, _p, _p],
    "mg_canny_nms": [_p, _i, _i, _i, _p, _p, _p, _p, _l, _p],
    "mg_canny_hysteresis": [_p, _p, _l, _i, _i, _i, _p, _p, _p, _p],
    "mg_unpack_bits": [_p, _l, _i, _l, _p, _p],
    "mg_hysteresis_tiles": [_i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mg_edge_grid_scan_words": [_i, _i, _i, _i],
    "mg_edge_grid": [_p, _l, _i, _i, _i, _i, _p, _p, _p, _p, _l, _p, _p, _i, _p],
    "mg_candidate_circles": [_p, _l, _p, _p, _p, _i, _i, _i, _i, _p, _l, _i, _i, _p, _l, _p, _p],
    "mg_candidate_keys": [_p, _l, _p, _p, _p, _i, _i, _i, _i, _p, _l, _i, _i, _p, _p, _p],
    "mg_keys_to_circles": [_p, _l, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _l, _p, _p, _p, _i, _p],
    "mg_bitmap_to_circles": [_p, _l, _i, _i, _i, _i, _i, _p, _p, _l, _p, _p],
    "mg_edge_angles": [_p, _i, _i, _i, _p, _l, _p, _p, _p],
    "mg_dedup_layout": [_i, _i, _i, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "mg_score_circles": [_p, _p, _p, _l, _i, _i, _i, _p, _l, _p, _p, _p, _i, _i, _p, _p, _p, _i, _f, _i, _i, _p, _p, _p, _p, _p, _p],
    "mg_score_keyed_supported": [_i, _i],
    "mg_score_pair_table": [_p, _i],
    "mg_score_pairs": [_i, _p, _i],
    "mg_score_circles_keyed": [_p, _p, _p, _p, _l, _i, _i, _i, _p, _l, _p, _p, _i, _i, _p, _p, _p, _i, _p, _f, _i,
                               _p, _p, _p, _p, _p, _p, _l, _p, _i, _p],
    "mg_nms_rounds": [_p, _l, _p, _p, _p, _p, _i, _i, _p, _i, _p, _l, _p, _p, _l, _i, _i, _p, _l, _p, _p],
    "mg_nms_same_centre": [_p, _l, _p, _p, _p, _p, _i, _i, _p, _l, _p, _p, _l, _p, _p],
    "mg_nms_sparse": [_p, _l, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p],
    "mg_nms_sparse_max_dist": [],
    "mg_collect_circles": [_p, _l, _p, _p, _p, _p, _i, _i, _p, _p, _l, _p, _p, _p, _i, _p],
    "mg_circle_labels": [_p, _l, _p, _i, _i, _i, _p, _i, _p, _i, _p],
    "mg_nms_cleanup": [_p, _l, _p, _p, _p, _p, _i, _i, _p, _i, _p, _l, _p, _l, _p, _p],
    "mg_roi_gather_reduce": [_p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "mg_roi_gather_reduce_batched": [_p, _i, _l, _i, _i, _i, _i, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "mg_counts_to_offsets": [_p, _i, _i, _p, _p],
    "mg_roi_segment_reduce": [_p, _i, _l, _i, _i, _i, _i, _i, _p, _l, _p, _i, _i, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p],
    "mg_roi_window_order": [_p, _l, _p, _i, _i, _p, _p],
    "mg_roi_masked_median": [_p, _i, _p, _l, _l, _i, _i, _i, _i, _p, _p],
    "mg_roi_masked_median_u16": [_p, _p, _i, _i, _i, _i, _p, _p],
    "mg_cluster1d_costs": [_p, _i, _i, _i, _d, _p, _d, _p, _p],
    "mg_button_masks": [_p, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p],
    "mg_stream_probe": [_p, _p, _l, _i, _p, _i, _p],
    "mg_masked_sums": [_p, _i, _p, _p, _i, _i, _i, _p, _p, _p],
    "mg_host_read_runs": [_p, _p, _p, _p, _i, _i, _p],
    "mg_host_write_runs": [_p, _p, _p, _p, _i, _i, _p],
}

RETURNS_INT64 = {"mg_scharr_hist_scratch_words", "mg_blur_hist_scratch_words", "mg_edge_grid_scan_words", "mg_flatfield_max_scratch_floats"}

_lib = None


class NativeLibraryMissing(RuntimeError):
    pass


def lib():
    """The loaded library; raises loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeLibraryMissing(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C magnify_amd/csrc).  magnify_amd has no CPU fallback."
            )
        handle = C.CDLL(LIB_PATH)
        for name, argtypes in PROTOTYPES.items():
            fn = getattr(handle, name)
            fn.argtypes = argtypes
            fn.restype = C.c_int64 if name in RETURNS_INT64 else C.c_int
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc == -1:
        raise ValueError(f"{what}: invalid argument (MG_EINVAL)")
    raise RuntimeError(f"{what}: HIP launch failure (code {rc})")


def dtype_code(dtype) -> int:
    """MG_* code of a numpy/torch dtype (by name)."""
    name = str(dtype).replace("torch.", "")
    try:
        return {"uint8": MG_U8, "uint16": MG_U16, "float32": MG_F32, "float64": MG_F64}[name]
    except KeyError:
        raise TypeError(f"unsupported image dtype {dtype}; supported: uint8, uint16, float32, float64") from None


# ---- host tables ---------------------------------------------------------------------------


def circle_points(r: int, four_connected: bool = False) -> np.ndarray:
    n = lib().mg_circle_points(int(r), int(bool(four_connected)), None, 0)
    if n < 0:
        raise ValueError("radius must be non-negative")
    out = np.empty((n, 2), dtype=np.int32)
    lib().mg_circle_points(int(r), int(bool(four_connected)), out.ctypes.data, n)
    return out


def disk_halfwidths(r: int) -> np.ndarray:
    out = np.empty(2 * r + 1, dtype=np.int32)
    if lib().mg_disk_halfwidths(int(r), out.ctypes.data) < 0:
        raise ValueError("filled disk is undefined for r < 2 (reference utils.py:398-430)")
    return out


def cv_disk_halfwidths(r: int) -> np.ndarray:
    out = np.empty(r + 1, dtype=np.int32)
    if lib().mg_cv_disk_halfwidths(int(r), out.ctypes.data) < 0:
        raise ValueError("radius must be non-negative")
    return out


def perimeter_table(min_r: int, max_r: int):
    starts = np.empty(max_r - min_r + 2, dtype=np.int32)
    total = lib().mg_perimeter_table(int(min_r), int(max_r), None, None, starts.ctypes.data, 0)
    if total < 0:
        raise ValueError("bad radius range")
    rc = np.empty((total, 2), dtype=np.int32)
    expected = np.empty(total, dtype=np.float64)
    lib().mg_perimeter_table(int(min_r), int(max_r), rc.ctypes.data, expected.ctypes.data, starts.ctypes.data, total)
    return rc, expected, starts


def score_pair_table():
    """Bound tables of mg_score_circles_keyed: uint64 (27, 80), 8 signed bytes per pair of opposite points."""
    total = lib().mg_score_pair_table(None, 0)
    entries = np.zeros(total, dtype=np.uint64)
    check(min(lib().mg_score_pair_table(entries.ctypes.data, total), 0), "mg_score_pair_table")
    return entries.reshape(27, 80)


def score_pairs(r: int):
    """First points (dr, dc) of the pairs of opposite perimeter points of radius r, in table order."""
    n = lib().mg_score_pairs(int(r), None, 0)
    out = np.empty((n, 2), dtype=np.int32)
    lib().mg_score_pairs(int(r), out.ctypes.data, n)
    return out


def dedup_layout(h: int, w: int, min_r: int, max_r: int):
    """(tile_rows, tile_cols, n_layers, bitmap_words) of the circle de-duplication bitmap."""
    a, b = C.c_int(0), C.c_int(0)
    n, words = C.c_int64(0), C.c_int64(0)
    check(lib().mg_dedup_layout(h, w, min_r, max_r, C.byref(a), C.byref(b), C.byref(n), C.byref(words)), "mg_dedup_layout")
    return a.value, b.value, n.value, words.value
