"""Host-side helpers with the reference's names (src/magnify/utils.py:55-99, 398-465).  The digital
circle tables come from the native library (no GPU needed)."""
from __future__ import annotations

import inspect
import os
import re
from collections.abc import Iterable

import numpy as np

from . import _native

PathLike = (str, bytes, os.PathLike)
_SEED = {"value": None, "counter": 0}


def seed(value):
    """Fix the RNG stream of the circle finders (the reference draws from an unseeded numba RNG,
    utils.py:309-320; by default this build draws a fresh seed from OS entropy per call)."""
    _SEED["value"] = None if value is None else int(value)
    _SEED["counter"] = 0


def next_seed() -> int:
    if _SEED["value"] is None:
        return int(np.random.SeedSequence().generate_state(2, dtype=np.uint32).view(np.uint64)[0])
    _SEED["counter"] += 1
    return (_SEED["value"] + 0x632BE59BD9B4E019 * _SEED["counter"]) & 0xFFFFFFFFFFFFFFFF


def ceildiv(a: int, b: int) -> int:
    return -(a // -b)


def bounding_box(x: int, y: int, box_length: int, image_width: int, image_height: int):
    """(top, bottom, left, right) of the L x L window clamped into the image (utils.py:60-80)."""
    def clamp(c, size):
        lo, hi = c - box_length // 2, c + ceildiv(box_length, 2)
        if lo < 0:
            lo, hi = 0, hi - lo
        if hi > size:
            lo, hi = lo - (hi - size), size
        return lo, hi

    top, bottom = clamp(y, image_height)
    left, right = clamp(x, image_width)
    return top, bottom, left, right


def roi_mark_chunk(bound: int, n_channels: int, n_times: int, roi_length: int, chunk_bytes: float = 1e6) -> int:
    """Markers per chunk of ``roi / fg / bg`` as the reference chunks them (find.py:185-188, 506-531): every channel
    and timepoint of a marker in one chunk, as many markers as make at least ``chunk_bytes`` -- counted in pixels, "since
    fg/bg bool arrays should also be 1MB" --, at most ``bound`` (the bead count; for chips the reference bounds by the
    number of marker ROWS, find.py:187)."""
    import math

    per_marker = roi_length ** 2 * n_channels * n_times
    return int(max(0, min(math.ceil(chunk_bytes / max(per_marker, 1)), bound)))


def circle_points(r, four_connected=False):
    return _native.circle_points(int(r), four_connected)


def filled_circle_points(r):
    r = int(r)
    hw = _native.disk_halfwidths(r)
    per = _native.circle_points(r)
    seen = {tuple(p) for p in per.tolist()}
    inner = [(dy, dx) for dy in range(-r, r + 1) for dx in range(-hw[dy + r], hw[dy + r] + 1) if (dy, dx) not in seen]
    return np.concatenate([per, np.asarray(inner, dtype=np.int32).reshape(-1, 2)])


def valid_kwargs(kwargs, func):
    args = list(inspect.signature(func).parameters)
    return {k: kwargs[k] for k in kwargs if k in args}


def natural_sort_key(s: str):
    return [int(t) if t.isdigit() else t.lower() for t in re.split("([0-9]+)", s)]


def to_list(x):
    if x is None:
        return []
    if not isinstance(x, Iterable) or isinstance(x, str):
        return [x]
    return list(x)
