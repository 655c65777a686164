"""Oracle: component-level restatement of the reference hot path on plain NumPy
arrays (the reference works on xarray/dask objects that cannot be imported here).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Arrays use the reference's
standardised layouts: tiles ``(channel, time, tile_row, tile_col, tile_y, tile_x)``
and images ``(channel, time, im_y, im_x)`` (preprocess.py:26, stitch.py:39).
"""
from __future__ import annotations

import math

import numpy as np

from . import ref_numeric as rn
from . import ref_opencv as cv

GRID_LENGTH = 20  # find.py:215, 348, 482

# --------------------------------------------------------------------------------------
# A1 stitch, A2 flat-field
# --------------------------------------------------------------------------------------


def stitch(tiles: np.ndarray, overlap: int) -> np.ndarray:
    """Crop overlap//2 (+ the odd remainder on the far side) from every tile edge
    and butt the tiles together; no blending (stitch.py:12-39)."""
    if overlap < 0:
        raise ValueError("Overlap must be non-negative.")
    c, t, nr, nc, ty, tx = tiles.shape
    if overlap >= ty or overlap >= tx:
        raise ValueError("Overlap must be smaller than tile size.")
    clip, rem = overlap // 2, overlap % 2
    crop = tiles[..., clip : ty - clip - rem, clip : tx - clip - rem]
    hy, hx = crop.shape[-2:]
    # (c, t, nr, nc, hy, hx) -> (c, t, nr, hy, nc, hx) -> (c, t, nr*hy, nc*hx)
    return np.ascontiguousarray(crop.transpose(0, 1, 2, 4, 3, 5)).reshape(c, t, nr * hy, nc * hx)


def flatfield_correct(tiles: np.ndarray, flatfield=1.0, darkfield=0.0) -> np.ndarray:
    """preprocess.py:83-87.  float64: clip(tile - dark, 0); M1 = global max; / flat;
    * M1 / (new global max); truncating cast back to the input dtype.
    PARITY UNPINNED: the reference has no test for this function."""
    t = np.clip(tiles.astype(np.float64) - darkfield, 0, None)
    m1 = t.max()
    t = t / flatfield
    with np.errstate(invalid="ignore", divide="ignore"):
        t = t * m1 / t.max()
        return t.astype(tiles.dtype)


# --------------------------------------------------------------------------------------
# find_circles (utils.py:102-222) with the RNG made explicit
# --------------------------------------------------------------------------------------


def edge_stage(img_u8: np.ndarray, low_edge_quantile: float, high_edge_quantile: float):
    """Steps 1-2 of find_circles (utils.py:115-142): blur, Scharr, quantile
    thresholds on the float32 gradient magnitude, Canny -> {0,1} edge map."""
    blur = cv.gaussian_blur5(img_u8)
    dx, dy = cv.scharr(blur)
    grad = np.sqrt(dx**2 + dy**2)
    lo = np.quantile(grad, low_edge_quantile)
    hi = np.quantile(grad, high_edge_quantile)
    edges = cv.canny(dx, dy, lo, hi)
    return blur, dx, dy, edges, (float(lo), float(hi))


def find_circles(img_u8, low_edge_quantile, high_edge_quantile, grid_length, num_iter, min_radius,
                 max_radius, min_roundness, min_dist, seed=0, picks=None, grad_angles=None):
    """utils.py:102-222.  ``seed`` selects the build's RNG stream; ``picks`` =
    (i0, j1, j2) overrides it with explicit draws."""
    _, dx, dy, edges, _ = edge_stage(img_u8, low_edge_quantile, high_edge_quantile)
    if picks is None:
        picks = rn.draw_picks(seed, num_iter, edges, grid_length)
    cand = rn.candidate_circles_from_picks(edges, grid_length, *picks)
    if len(cand) == 0:
        return np.empty((0, 3), dtype=np.int32), np.empty(0, dtype=np.float32)
    return rn.filter_circles(cand, edges, dx, dy, min_radius, max_radius, min_roundness, min_dist,
                             grad_angles=grad_angles)


# --------------------------------------------------------------------------------------
# A12 BeadFinder, A18 ROI reduce
# --------------------------------------------------------------------------------------


def bead_params(min_bead_diameter, max_bead_diameter, roi_length=None):
    """find.py:458-467."""
    if min_bead_diameter > max_bead_diameter:
        raise ValueError("min_bead_diameter must be <= max_bead_diameter.")
    min_r = math.floor(min_bead_diameter / 2)
    max_r = math.ceil(max_bead_diameter / 2)
    length = roi_length if roi_length is not None else 2 * max_bead_diameter
    return min_r, max_r, length


def dedup_against(seen_rc: np.ndarray, new: np.ndarray, radius: float) -> np.ndarray:
    """find.py:490-500: drop rows of ``new`` that have any earlier bead within
    ``radius`` (Euclidean, inclusive -- KDTree.query_ball_point semantics)."""
    if len(seen_rc) == 0 or len(new) == 0:
        return new
    d2 = ((new[:, None, :2].astype(np.float64) - seen_rc[None, :, :2].astype(np.float64)) ** 2).sum(-1)
    return new[~(d2 <= float(radius) ** 2).any(axis=1)]


def find_beads(image: np.ndarray, min_bead_diameter, max_bead_diameter, low_edge_quantile=0.1,
               high_edge_quantile=0.9, num_iter=5000000, min_roundness=0.3, roi_length=None,
               search_channels=None, seed=0):
    """BeadFinder.__call__ (find.py:471-605) on an ``image (C, T, H, W)``.

    Detection runs on time index 0 of every search channel; geometry is replicated
    over time.  Returns a dict with the reference's variables: roi (M,C,T,L,L),
    fg/bg (M,T,L,L) bool, x/y (M,T) float64, valid (M,T) bool, plus the bead
    table ``beads (M,3)`` (row, col, r)."""
    min_r, max_r, length = bead_params(min_bead_diameter, max_bead_diameter, roi_length)
    n_c, n_t, h, w = image.shape
    if search_channels is None:
        search_channels = list(range(n_c))
    beads = np.empty((0, 3))
    for k, ch in enumerate(search_channels):
        u8 = rn.to_uint8(image[ch, 0])
        b, _ = find_circles(u8, low_edge_quantile, high_edge_quantile, GRID_LENGTH, num_iter, min_r, max_r,
                            min_roundness, min_dist=min_r, seed=seed + k)
        b = dedup_against(beads, b, 2 * min_r)
        beads = np.concatenate([beads, b])
    m = len(beads)
    out = {
        "beads": beads.astype(np.int64),
        "x": np.repeat(beads[:, None, 1], n_t, axis=1).astype(np.float64),
        "y": np.repeat(beads[:, None, 0], n_t, axis=1).astype(np.float64),
        "valid": np.ones((m, n_t), dtype=bool),
        "roi": np.zeros((m, n_c, n_t, length, length), dtype=image.dtype),
        "fg": np.zeros((m, n_t, length, length), dtype=bool),
        "bg": np.zeros((m, n_t, length, length), dtype=bool),
    }
    if m == 0:
        return out
    labels = rn.circle_labels(beads.astype(int), h, w)
    for i in range(m):
        top, bottom, left, right = rn.bounding_box(round(out["x"][i, 0]), round(out["y"][i, 0]), length, w, h)
        sub = labels[top:bottom, left:right]
        out["fg"][i] = (sub == i)[None]
        out["bg"][i] = (sub == -1)[None]
        out["roi"][i] = image[:, :, top:bottom, left:right]
    return out


def roi_reduce(roi: np.ndarray, fg: np.ndarray, bg: np.ndarray, medians: bool = True):
    """The ROI reductions of README.md:21-22, identify.py:76-80, filter.py:21-22.

    roi (M,C,T,L,L), masks (M,T,L,L).  Integer sums and counts are exact;
    mean = sum / count in float64 (NaN for an empty mask, as xarray's nanmean);
    median = numpy nanmedian semantics (mean of the two middle values).
    PARITY UNPINNED: no reference test covers these expressions."""
    f = fg[:, None].astype(bool)
    b = bg[:, None].astype(bool)
    wide = roi.astype(np.float64) if roi.dtype.kind == "f" else roi.astype(np.int64)
    res = {
        "fg_count": fg.sum(axis=(-1, -2)).astype(np.int64),
        "bg_count": bg.sum(axis=(-1, -2)).astype(np.int64),
        "fg_sum": np.where(f, wide, 0).sum(axis=(-1, -2)),
        "bg_sum": np.where(b, wide, 0).sum(axis=(-1, -2)),
    }
    with np.errstate(invalid="ignore", divide="ignore"):
        res["fg_mean"] = res["fg_sum"] / res["fg_count"][:, None].astype(np.float64)
        res["bg_mean"] = res["bg_sum"] / res["bg_count"][:, None].astype(np.float64)
    if not medians:
        return res
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r64 = roi.astype(np.float64)
        shp = r64.shape[:3] + (-1,)
        res["fg_median"] = np.nanmedian(np.where(f, r64, np.nan).reshape(shp), axis=-1)
        res["bg_median"] = np.nanmedian(np.where(b, r64, np.nan).reshape(shp), axis=-1)
    return res


# --------------------------------------------------------------------------------------
# A16 grid fit for chips: cluster_1d, label_clusters, regress_clusters
# --------------------------------------------------------------------------------------


def cluster_1d(points, total_length, num_clusters, cluster_length, ideal_num_points, penalty):
    """Brute-force search over integer offsets for the best run of equal-width
    clusters (find.py:632-677).  First minimum wins (strict <)."""
    points = np.asarray(points, dtype=np.float64)
    ideal = np.asarray(ideal_num_points)
    perm = np.argsort(points)
    pts = points[perm]
    best_cost, best_spans = np.inf, None
    for offset in range(total_length - round(num_clusters * cluster_length)):
        bounds = np.arange(num_clusters + 1) * cluster_length + offset
        centers = (bounds[1:] + bounds[:-1]) / 2
        spans = np.searchsorted(pts, bounds)
        n_in = spans[1:] - spans[:-1]
        sq = (pts[spans[0] : spans[-1]] - np.repeat(centers, n_in)) ** 2
        run = np.insert(np.cumsum(sq), 0, 0)
        cost = np.diff(run[spans - spans[0]])
        has = n_in > 0
        cost[has] /= n_in[has]
        cost[~has] = np.max(cost)
        cost *= np.sqrt(ideal)
        cost = cost + penalty * (ideal - n_in) ** 2
        total = cost.sum()
        if total < best_cost:
            best_cost, best_spans = total, spans
    labels = -np.ones_like(pts, dtype=int)
    labels[best_spans[0] : best_spans[-1]] = np.repeat(np.arange(num_clusters), best_spans[1:] - best_spans[:-1])
    return labels[np.argsort(perm)]


def label_clusters(points, offset, num_clusters, cluster_length, cluster_gap):
    """Fixed-offset labelling when the chip's top/left edge is given (find.py:680-695)."""
    points = np.asarray(points, dtype=np.float64)
    perm = np.argsort(points)
    pts = points[perm]
    labels = -np.ones_like(pts, dtype=int)
    steps = [offset] + ([cluster_length, cluster_gap] * num_clusters)[:-1]
    spans = np.searchsorted(pts, np.cumsum(steps))
    for i in range(num_clusters):
        labels[spans[2 * i] : spans[2 * i + 1]] = i
    return labels[np.argsort(perm)]


def _linregress(x, y):
    """slope, intercept of scipy.stats.linregress (covariance form:
    slope = ssxym / ssxm, intercept = ymean - slope * xmean)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    xm, ym = np.mean(x), np.mean(y)
    ssxm, ssxym, _, _ = np.cov(x, y, bias=1).flat
    slope = ssxym / ssxm
    return slope, ym - slope * xm


def regress_clusters(x, y, labels, num_clusters, ideal_num_points):
    """Lines through each cluster with a shared (median) slope and intercepts
    blended with a global evenly-spaced estimate (find.py:698-748)."""
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if num_clusters == 1:
        if len(x) == 1:
            return 0, y
        return _linregress(x, y)
    slopes = np.full(num_clusters, np.nan)
    intercepts = np.full(num_clusters, np.nan)
    groups = [(x[labels == i], y[labels == i]) for i in range(num_clusters)]
    for i, (gx, gy) in enumerate(groups):
        if len(gx) > 1:
            slopes[i], intercepts[i] = _linregress(gx, gy)
    slope = np.nanmedian(slopes)
    for i, (gx, gy) in enumerate(groups):
        if len(gx) > 0:
            intercepts[i] = np.median(gy - slope * gx)
    ok = ~np.isnan(intercepts)
    idx = np.arange(num_clusters)
    g_m, g_b = _linregress(idx[ok], intercepts[ok])
    for i, (gx, _) in enumerate(groups):
        if ideal_num_points[i] != 0 and ok[i]:
            wgt = min(len(gx), ideal_num_points[i]) / ideal_num_points[i]
            intercepts[i] = wgt * intercepts[i] + (1 - wgt) * (g_m * i + g_b)
        else:
            intercepts[i] = g_m * i + g_b
    return slope, intercepts


# --------------------------------------------------------------------------------------
# A16/A17 ButtonFinder
# --------------------------------------------------------------------------------------


def button_params(min_button_diameter, max_button_diameter, chamber_diameter, roi_length=None):
    """find.py:34-49."""
    if min_button_diameter > max_button_diameter:
        raise ValueError("min_button_diameter must be <= max_button_diameter.")
    return (math.floor(min_button_diameter / 2), math.ceil(max_button_diameter / 2), round(chamber_diameter / 2),
            roi_length if roi_length is not None else round(1.2 * chamber_diameter))


def find_centers(images, tag, row_dist, col_dist, min_r, max_r, chamber_r, low_q, high_q, num_iter,
                 min_roundness, cluster_penalty, top_chamber=None, left_chamber=None, seed=0):
    """ButtonFinder.find_centers (find.py:205-306).  ``images`` is (n_search, H, W)."""
    points = np.empty((0, 2))
    for k, image in enumerate(images):
        u8 = rn.to_uint8(image)
        new, _ = find_circles(u8, low_q, high_q, GRID_LENGTH, num_iter, min_r, max_r, min_roundness,
                              min_dist=chamber_r, seed=seed + k)
        new = new[:, :2]
        if len(points) > 0:
            dist = np.linalg.norm(points[np.newaxis] - new[:, np.newaxis], axis=2)
            new = new[np.min(dist, axis=1) > chamber_r]
        points = np.concatenate([points, new])
    x, y = points[:, 1], points[:, 0]
    per_row = (tag != "").sum(axis=1)
    per_col = (tag != "").sum(axis=0)
    n_rows, n_cols = tag.shape
    h, w = images[0].shape
    if top_chamber is None:
        row_labels = cluster_1d(y, h, n_rows, row_dist, per_row, cluster_penalty)
    else:
        row_labels = label_clusters(y, top_chamber, n_rows, 2 * chamber_r, row_dist - 2 * chamber_r)
    if left_chamber is None:
        col_labels = cluster_1d(x, w, n_cols, col_dist, per_col, cluster_penalty)
    else:
        col_labels = label_clusters(x, left_chamber, n_cols, 2 * chamber_r, col_dist - 2 * chamber_r)
    inside = (row_labels >= 0) & (col_labels >= 0)
    x, y, row_labels, col_labels = x[inside], y[inside], row_labels[inside], col_labels[inside]
    row_slope, row_b = regress_clusters(x, y, row_labels, n_rows, per_row)
    col_slope, col_b = regress_clusters(y, x, col_labels, n_cols, per_col)
    row_b, col_b = np.atleast_1d(row_b), np.atleast_1d(col_b)
    mark_y = (row_slope * col_b[np.newaxis] + row_b[:, np.newaxis]) / (1 - row_slope * col_slope)
    mark_x = mark_y * col_slope + col_b[np.newaxis]
    return mark_x, mark_y


def find_rois(images, x, y, tag, search_idx, min_r, max_r, chamber_r, length, low_q, num_iter, min_roundness,
              seed=0):
    """ButtonFinder.find_rois (find.py:308-402) for one timestep.  ``images`` (C,H,W)."""
    n_rows, n_cols = tag.shape
    n_c, h, w = images.shape
    x, y = x.copy(), y.copy()
    roi = np.empty((n_rows, n_cols, n_c, length, length), dtype=images.dtype)
    fg = np.empty((n_rows, n_cols, length, length), dtype=bool)
    bg = np.empty_like(fg)
    hi_q = 1 - np.pi * min_r / length**2
    per_chamber_iter = num_iter // (n_rows * n_cols)
    for i in range(n_rows):
        for j in range(n_cols):
            top, bottom, left, right = rn.bounding_box(round(x[i, j]), round(y[i, j]), length, w, h)
            roi[i, j] = images[:, top:bottom, left:right]
            best, best_score = None, -np.inf
            if tag[i, j] != "":
                for k, ch in enumerate(search_idx):
                    sub = rn.to_uint8(roi[i, j, ch])
                    circles, scores = find_circles(sub, low_q, hi_q, GRID_LENGTH, per_chamber_iter, min_r, max_r,
                                                   min_roundness, min_dist=0,
                                                   seed=chamber_seed(seed, i * n_cols + j, k))
                    if len(circles) > 0:
                        idx = np.argmax(scores)
                        if scores[idx] > best_score:
                            best, best_score = circles[idx], scores[idx]
            radius = max_r
            if best is not None:
                y[i, j], x[i, j] = best[:2]
                x[i, j] += left
                y[i, j] += top
                top, bottom, left, right = rn.bounding_box(round(x[i, j]), round(y[i, j]), length, w, h)
                roi[i, j] = images[:, top:bottom, left:right]
                radius = int(best[2])
            x_rel, y_rel = round(x[i, j]) - left, round(y[i, j]) - top
            outer = cv.filled_circle_mask((length, length), (y_rel, x_rel), chamber_r)
            inner = cv.filled_circle_mask((length, length), (y_rel, x_rel), max_r)
            bg[i, j] = outer & ~inner
            fg[i, j] = cv.filled_circle_mask((length, length), (y_rel, x_rel), radius)
    return roi, fg, bg, x, y


def chamber_seed(seed: int, chamber: int, k: int) -> int:
    """RNG stream id of the per-chamber refinement (build-defined)."""
    return (seed + 0x51ED270B * (chamber + 1) + 0x2545F491 * (k + 1)) & 0xFFFFFFFFFFFFFFFF
