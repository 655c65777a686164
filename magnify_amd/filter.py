"""Marker filters on the device reductions (reference src/magnify/filter.py).

``filter_expression`` (filter.py:11-37) and ``filter_leaky`` (filter.py:65-94) only need the masked
medians of the first timestep, which ``mg_roi_masked_median`` computes on the gathered ROIs; the
thresholding is the reference's own NumPy expression, evaluated on the (mark,) vectors on the host.
``filter_nonround`` (filter.py:40-62) measures contours with cv.findContours / cv.arcLength: restated here
as Moore border tracing of every connected component of the small fg masks, on the host.
"""
from __future__ import annotations

import numpy as np

from . import reduce, registry
from .utils import to_list


def _first_step_medians(assay):
    """fg and bg median of every (mark, channel) at time 0 as float64 numpy arrays: ``assay.isel(time=0)`` first,
    then the medians (filter.py:20-22, 69-75) -- the masks of time 0, whatever the later timepoints hold."""
    fg = reduce.masked_median(assay, "fg", time=0).data[:, :, 0].cpu().numpy()
    bg = reduce.masked_median(assay, "bg", time=0).data[:, :, 0].cpu().numpy()
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


# 8-neighbourhood in clockwise order starting east (row, col steps) and the step lengths along it
_RING = np.array([(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)])
_STEP = np.array([1.0, np.sqrt(2.0)] * 4)


def outer_border_length(blob: np.ndarray) -> float:
    """Length of the closed outer border of ONE 8-connected component (``blob`` bool, component only),
    followed through its border pixels' centres: what cv.findContours(RETR_EXTERNAL) + cv.arcLength(closed)
    measure (collapsing straight runs, CHAIN_APPROX_SIMPLE, does not change the length).  Moore-neighbour
    tracing from the component's first pixel in raster order, stopping when the start pixel is entered again
    in the start direction.  A single pixel has length 0; a one-pixel-wide line is walked there and back."""
    h, w = blob.shape
    pad = np.zeros((h + 2, w + 2), dtype=bool)
    pad[1:-1, 1:-1] = blob
    ys, xs = np.nonzero(pad)
    start = (int(ys[0]), int(xs[0]))  # raster-first pixel: its west and north neighbours are background
    cur, came = start, 4              # pretend we arrived from the west
    first_move = None
    length = 0.0
    for _ in range(4 * pad.size + 8):
        for turn in range(1, 9):      # clockwise from the neighbour after the one we came from
            d = (came + turn) % 8
            nxt = (cur[0] + _RING[d][0], cur[1] + _RING[d][1])
            if pad[nxt]:
                break
        else:
            return 0.0                # isolated pixel
        if cur == start and first_move is not None and d == first_move:
            return length
        if first_move is None:
            first_move = d
        length += _STEP[d]
        cur, came = nxt, (d + 4) % 8
    raise RuntimeError("border following did not close")


def mask_perimeter(mask: np.ndarray) -> float:
    """Total outer-border length of the 8-connected components of ``mask`` that do not lie inside a hole of another
    component (filter.py:51-52: cv.findContours(..., RETR_EXTERNAL, ...) + cv.arcLength)."""
    import scipy.ndimage

    mask = np.asarray(mask, dtype=bool)
    labels, n = scipy.ndimage.label(mask, structure=np.ones((3, 3), dtype=bool))
    if n == 0:
        return 0.0
    # background connected to the frame (4-connected, the dual of the 8-connected foreground)
    pad = np.zeros((mask.shape[0] + 2, mask.shape[1] + 2), dtype=bool)
    pad[1:-1, 1:-1] = mask
    bg, _ = scipy.ndimage.label(~pad)
    outside = bg == bg[0, 0]
    total = 0.0
    firsts = scipy.ndimage.find_objects(labels)
    for k in range(1, n + 1):
        sl = firsts[k - 1]
        comp = labels[sl] == k
        # the pixel above the component's raster-first pixel belongs to the background region around its outer border
        y0 = sl[0].start
        x0 = sl[1].start + int(np.argmax(comp[0]))
        if outside[y0, x0 + 1]:  # (padded coordinates: row y0 - 1 + 1, column x0 + 1)
            total += outer_border_length(comp)
    return total


@registry.component("filter_nonround")
def filter_nonround(assay, min_roundness=0.75, search_channel=None):
    """filter.py:40-62: a marker stays valid only if its foreground mask (time 0) is round enough,
    4 pi area / perimeter^2 > min_roundness, perimeter = total length of the outer borders of its connected
    components; a mask without any border length (empty or single pixels) is invalid.  The masks are a few
    thousand small images per assay: traced on the host.  Components that lie inside a hole of another one are
    skipped, as RETR_EXTERNAL skips them.  PARITY UNPINNED against OpenCV itself (not available here); checked against
    an independent restatement of its border following (oracle/ref_contours.py, tests/test_cpu_host.py)."""
    import scipy.ndimage

    _channel_indexes(assay, search_channel)  # validates the names like the other filters
    fg = assay.coords["fg"] if "fg" in assay.coords else assay.data_vars["fg"]
    masks = fg.transpose("mark", "time", "roi_y", "roi_x").data
    masks = (masks[:, 0].cpu().numpy() if hasattr(masks, "cpu") else np.asarray(masks)[:, 0]).astype(bool)
    var, valid = _valid_array(assay)
    flat = valid.reshape(len(masks), -1)
    for i, mask in enumerate(masks):
        perimeter = mask_perimeter(mask)
        if perimeter == 0:
            flat[i] = False
            continue
        flat[i] &= 4 * np.pi * float(mask.sum()) / perimeter**2 > min_roundness
    return assay.assign_coords(valid=(var.dims, flat.reshape(valid.shape)))
