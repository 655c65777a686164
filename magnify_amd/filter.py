"""Marker filters on the device reductions (reference src/magnify/filter.py).

``filter_expression`` (filter.py:11-37) and ``filter_leaky`` (filter.py:65-94) only need the masked
medians of the first timestep, which ``mg_roi_masked_median_u16`` computes on the gathered ROIs; the
thresholding is the reference's own NumPy expression, evaluated on the (mark,) vectors on the host.
``filter_nonround`` (filter.py:40-62) measures contours with cv.findContours / cv.arcLength and is
not part of this build.
"""
from __future__ import annotations

import numpy as np

from . import reduce, registry
from .utils import to_list


def _first_step_medians(assay):
    """fg and bg median of every (mark, channel) at time 0 as float64 numpy arrays."""
    fg = reduce.masked_median(assay, "fg").transpose("mark", "channel", "time").data[:, :, 0].cpu().numpy()
    bg = reduce.masked_median(assay, "bg").transpose("mark", "channel", "time").data[:, :, 0].cpu().numpy()
    return fg, bg


def _channel_indexes(assay, search_channel):
    names = np.asarray(assay.coords["channel"].values).tolist() if "channel" in assay.coords else None
    if search_channel is None:
        return list(range(assay.sizes["channel"]))
    wanted = to_list(search_channel)
    if names is None:
        return [int(c) for c in wanted]
    return [names.index(c) for c in wanted]


def _pairwise_spread(bg: np.ndarray) -> float:
    """std of the differences between every ordered pair of distinct background medians
    (filter.py:23-27, 77-80)."""
    d = bg[:, np.newaxis] - bg[np.newaxis, :]
    return float(d[~np.eye(len(bg), dtype=bool)].std())


def _valid_array(assay):
    v = assay.coords["valid"] if "valid" in assay.coords else assay.data_vars["valid"]
    return v, np.array(v.values, dtype=bool)


@registry.component("filter_expression")
def filter_expression(assay, search_channel=None, min_contrast=None):
    """Keep markers whose fg median exceeds the bg median by ``min_contrast`` (default: 4 standard
    deviations of the pairwise background differences) in at least one search channel."""
    fg, bg = _first_step_medians(assay)
    keep = np.zeros(assay.sizes["mark"], dtype=bool)
    for c in _channel_indexes(assay, search_channel):
        bound = 4 * _pairwise_spread(bg[:, c]) if min_contrast is None else min_contrast
        keep |= (fg[:, c] - bg[:, c]) > bound
    var, valid = _valid_array(assay)
    valid &= keep.reshape((-1,) + (1,) * (valid.ndim - 1)) if var.dims[0] == "mark" else keep
    return assay.assign_coords(valid=(var.dims, valid))


@registry.component("filter_leaky")
def filter_leaky_buttons(assay, search_channel=None):
    """A tagged button next (in the flattened mark order, same chip row side) to an untagged one is
    kept only while that blank neighbour shows no expression (fg - bg below 5 sigma)."""
    fg, bg = _first_step_medians(assay)
    tag = np.asarray(assay.coords["tag"].values).reshape(-1)
    rows = np.asarray(assay.coords["mark_row"].values).reshape(-1)
    var, valid = _valid_array(assay)
    flat = valid.reshape(len(tag), -1)
    last_row = rows.max()
    for c in _channel_indexes(assay, search_channel):
        empty = (fg[:, c] - bg[:, c]) < 5 * _pairwise_spread(bg[:, c])
        for i in range(len(tag)):
            if tag[i] == "":
                continue
            if rows[i] > 0 and tag[i - 1] == "":
                flat[i] &= empty[i - 1]
            if rows[i] < last_row and tag[i + 1] == "":
                flat[i] &= empty[i + 1]
    return assay.assign_coords(valid=(var.dims, flat.reshape(valid.shape)))


@registry.component("filter_nonround")
def filter_nonround(assay, min_roundness=0.75, search_channel=None):
    raise NotImplementedError("filter_nonround needs OpenCV's contour tracing (cv.findContours / cv.arcLength); "
                              "it is outside this build")
