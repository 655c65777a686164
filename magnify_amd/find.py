"""BeadFinder / ButtonFinder components (reference: src/magnify/find.py).

The numeric work runs in the HIP kernels behind ``hotpath``; this file keeps the reference's
constructor arguments, validation, output variables and orchestration order."""
from __future__ import annotations

import math

import numpy as np
import torch

from . import hotpath, preprocess, registry, utils
from .stack import dedup_against
from .xr_lite import DataArray

_FINDERS = {}


def _finder(n_planes, h, w, min_r, max_r, num_iter, device):
    """Workspace cache: one CircleFinder per problem shape."""
    key = (n_planes, h, w, min_r, max_r, num_iter, str(device))
    if key not in _FINDERS:
        if len(_FINDERS) > 4:
            _FINDERS.clear()
        _FINDERS[key] = hotpath.CircleFinder(n_planes, h, w, min_r, max_r, num_iter, device=device)
    return _FINDERS[key]


def _channel_index(assay, channel):
    labels = assay.coords["channel"].values.tolist() if "channel" in assay.coords else None
    if labels is not None and channel in labels:
        return labels.index(channel)
    if isinstance(channel, (int, np.integer)) and 0 <= int(channel) < assay.sizes["channel"]:
        return int(channel)
    raise KeyError(f"channel {channel!r} not found")


def _image_tensor(assay):
    img = assay.data_vars["image"].transpose("channel", "time", "im_y", "im_x")
    data = img.data
    if not isinstance(data, torch.Tensor) or not data.is_cuda:
        data = preprocess.to_device(data)
        assay["image"] = DataArray(data, ("channel", "time", "im_y", "im_x"))
    return data.contiguous()


def _plane_minmax(assay, image, idx):
    """(1, 2) float64 min/max of image plane ``idx`` = (channel, time); reuses what stitch computed."""
    c, t = image.shape[:2]
    cached = assay._cache.get("image_minmax")
    if cached is not None and cached[0] == image.data_ptr() and cached[1] is not None:
        return cached[1].view(c, t, 2)[idx[0], idx[1]].reshape(1, 2).contiguous()
    return None


class BeadFinder:
    def __init__(self, min_bead_diameter, max_bead_diameter, low_edge_quantile, high_edge_quantile, num_iter,
                 min_roundness, roi_length, search_channel, interactive):
        if min_bead_diameter > max_bead_diameter:
            raise ValueError("min_bead_diameter must be <= max_bead_diameter.")
        if interactive:
            raise NotImplementedError("the napari tuning GUI is outside the MI355X hot path")
        self.min_bead_radius = math.floor(min_bead_diameter / 2)
        self.max_bead_radius = math.ceil(max_bead_diameter / 2)
        self.low_edge_quantile = low_edge_quantile
        self.high_edge_quantile = high_edge_quantile
        self.num_iter = num_iter
        self.min_roundness = min_roundness
        self.roi_length = roi_length if roi_length is not None else 2 * max_bead_diameter
        self.search_channels = utils.to_list(search_channel)

    def __call__(self, assay):
        """find.py:471-605."""
        image = _image_tensor(assay)
        n_c, n_t, h, w = image.shape
        if not self.search_channels:
            self.search_channels = (assay.coords["channel"].values.tolist() if "channel" in assay.coords
                                    else list(range(n_c)))
        beads = np.empty((0, 3), dtype=np.int32)
        finder = _finder(1, h, w, self.min_bead_radius, self.max_bead_radius, self.num_iter, image.device)
        m, L = None, self.roi_length
        if len(self.search_channels) == 1:
            # one search channel (nothing to de-duplicate across channels, find.py:490-500): the ROI pass reads the
            # bead table where the suppression left it on the device; the host copy of the table runs beside it
            ch = _channel_index(assay, self.search_channels[0])
            roi_pass = lambda d_tab, d_num, cap: hotpath.roi_gather_reduce(  # noqa: E731  (queued before the counts come back)
                image[None], None, L, None, disks=True, device_tables=(d_tab, None, self.max_bead_radius),
                device_counts=(d_num, cap, None))
            counts, (d_out, d_scores, _) = finder.find(
                image[ch, 0:1], _plane_minmax(assay, image, (ch, 0)), self.low_edge_quantile, self.high_edge_quantile,
                self.min_roundness, self.min_bead_radius, [utils.next_seed()], host_results=False, follow=roi_pass)
            out = hotpath.finish_roi(finder.follow_result, counts)  # (launched for the full capacity: never None)
            beads = finder.fetch_results(counts, d_out, d_scores, overlap=True)[0][0]
        else:
            for channel in self.search_channels:
                ch = _channel_index(assay, channel)
                res, _ = finder.find(image[ch, 0:1], _plane_minmax(assay, image, (ch, 0)), self.low_edge_quantile,
                                     self.high_edge_quantile, self.min_roundness, self.min_bead_radius,
                                     [utils.next_seed()])
                b = dedup_against(beads, res[0][0], 2 * self.min_bead_radius)  # find.py:490-500
                beads = np.concatenate([beads, b])
            # masks straight from the bead table (what utils.circle_labels + the == i / == -1 tests yield,
            # find.py:561-586) -- no label map is written or read
            out = hotpath.roi_gather_reduce(image[None], [beads], L, None, disks=True)
        m = len(beads)
        # the kernel writes 0 / 1 bytes: reinterpreted as bool, not converted (two passes over M L^2 bytes less)
        fg = out["fg"].view(torch.bool)[:, None].expand(m, n_t, L, L)  # geometry replicated over time (find.py:585-586)
        bg = out["bg"].view(torch.bool)[:, None].expand(m, n_t, L, L)
        # the reference's chunk policy (find.py:506-531): every channel / timepoint of a marker together, markers per
        # chunk for >= 1 MB; recorded on the variables (xr_lite.DataArray.chunk: what to_xarray / mg.save go by)
        per = {"mark": utils.roi_mark_chunk(m, n_c, n_t, L)}
        assay["roi"] = DataArray(out["roi"], ("mark", "channel", "time", "roi_y", "roi_x")).chunk(per)
        xy = beads.astype(np.float64)
        assay = assay.assign_coords(
            fg=DataArray(fg, ("mark", "time", "roi_y", "roi_x")).chunk(per),
            bg=DataArray(bg, ("mark", "time", "roi_y", "roi_x")).chunk(per),
            x=(("mark", "time"), np.repeat(xy[:, None, 1], n_t, axis=1)),
            y=(("mark", "time"), np.repeat(xy[:, None, 0], n_t, axis=1)),
            valid=(("mark", "time"), np.ones((m, n_t), dtype=bool)),
        )
        # extras (not in the reference's schema): the fused masked reductions and the bead radii
        assay._cache["roi_sums"] = out["sums"]      # (mark, channel, time, {fg, bg}) float64
        assay._cache["roi_counts"] = out["counts"]  # (mark, {fg, bg}) int32
        assay._cache["radius"] = beads[:, 2].copy()
        return assay

    @registry.components.register("find_beads")
    def make(min_bead_diameter, max_bead_diameter, low_edge_quantile, high_edge_quantile, num_iter, min_roundness,
             roi_length, search_channel, interactive):
        return BeadFinder(min_bead_diameter=min_bead_diameter, max_bead_diameter=max_bead_diameter,
                          low_edge_quantile=low_edge_quantile, high_edge_quantile=high_edge_quantile,
                          num_iter=num_iter, min_roundness=min_roundness, roi_length=roi_length,
                          search_channel=search_channel, interactive=interactive)


# --------------------------------------------------------------------------------------
# ButtonFinder (find.py:13-442)
# --------------------------------------------------------------------------------------


def _group_stats(labels, num_clusters, *columns):
    """Per-cluster counts and means of the given columns (rows with label < 0 belong to no cluster)."""
    inside = labels >= 0
    lab = labels[inside]
    n = np.bincount(lab, minlength=num_clusters).astype(np.float64)
    with np.errstate(invalid="ignore", divide="ignore"):
        means = [np.bincount(lab, weights=c[inside], minlength=num_clusters) / n for c in columns]
    return inside, lab, n, means


def _line_fit(x, y):
    """Slope and intercept as scipy.stats.linregress computes them (find.py:710, 719, 735), operation for operation:
    the means by np.mean, the mean squares by np.cov(x, y, bias=1) -- centred rows, one matrix product, times 1 / n --,
    slope = ssxym / ssxm, intercept = ymean - slope * xmean.  (A grouped-sums variant, Sxy / Sxx over bincounts, agreed
    to an ulp or two; the slopes feed a median and rounded chamber centres, so the product does what the reference's
    library does.)"""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    if len(x) > 1 and np.amax(x) == np.amin(x):
        raise ValueError("Cannot calculate a linear regression if all x values are identical")  # (scipy's own refusal)
    xmean, ymean = np.mean(x, None), np.mean(y, None)
    ssxm, ssxym, _, _ = np.cov(x, y, bias=1).flat
    slope = ssxym / ssxm
    return slope, ymean - slope * xmean


def _cluster_slopes(gx, gy, order0, bounds, cnt, num_clusters):
    """``_line_fit(...)[0]`` of every cluster with two points or more.  Clusters of the same size are fitted together
    with the same operations on stacked arrays: row means of a (k, 2, n) block (NumPy's pairwise row sum, as np.mean
    of each row alone), the centred block times its transpose through np.matmul (one dgemm per cluster, as np.cov's
    np.dot), times 1 / n.  tests/test_oracle_golden.py compares it bit for bit with the per-cluster np.cov form and
    with scipy's linregress on reference-generated goldens.  (A chip has a few distinct sizes -- 28 buttons a row,
    one or two missed here and there: a few groups instead of 56 Python-level fits, 1 ms of a 4 ms C3 call.)"""
    slopes = np.full(num_clusters, np.nan)
    for size in np.unique(cnt[cnt >= 2]):
        which = np.flatnonzero(cnt == size)
        sel = order0[bounds[which][:, None] + np.arange(size)[None, :]]  # (k, size): the points in their original order
        X = np.empty((len(which), 2, size))
        X[:, 0], X[:, 1] = gx[sel], gy[sel]
        if (X[:, 0].max(axis=1) == X[:, 0].min(axis=1)).any():
            raise ValueError("Cannot calculate a linear regression if all x values are identical")  # (scipy's refusal)
        X -= X.mean(axis=2)[:, :, None]
        c = np.matmul(X, X.transpose(0, 2, 1).conj())
        c *= np.true_divide(1, size)
        slopes[which] = c[:, 0, 1] / c[:, 0, 0]
    return slopes


def label_clusters(points, offset, num_clusters, cluster_length, cluster_gap):
    """find.py:680-695: cluster i is the interval [offset + i (length + gap), ... + length); -1 outside.
    One binary search of every point in the 2 n interval ends (no sort of the points, no loop over clusters)."""
    points = np.asarray(points, dtype=np.float64)
    ends = np.cumsum(np.concatenate([[offset], np.tile([cluster_length, cluster_gap], num_clusters)[:-1]]))  # as summed there
    k = np.searchsorted(ends, points, side="right") - 1  # the last end <= point
    return np.where((k >= 0) & (k % 2 == 0), k // 2, -1).astype(int)


def regress_clusters(x, y, labels, num_clusters, ideal_num_points):
    """find.py:698-748: a line per cluster, the median of their slopes for all, per-cluster median intercepts at
    that slope, blended -- by how complete a cluster is -- with the evenly spaced estimate from a line through
    (cluster index, intercept).  The fits are linregress's own operations (_line_fit; clusters of equal size fitted
    together, _cluster_slopes); the medians of all clusters come from one sort."""
    x, y = np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64)
    labels = np.asarray(labels)
    if num_clusters == 1:
        return (0, y) if len(x) == 1 else _line_fit(x, y)
    ideal = np.asarray(ideal_num_points, dtype=np.float64)
    inside, lab, n, _ = _group_stats(labels, num_clusters)
    gx, gy = x[inside], y[inside]
    # a line per cluster with two points or more (the cluster's points in their original order, as x[labels == i])
    order0 = np.argsort(lab, kind="stable")
    bounds = np.concatenate([[0], np.cumsum(n)]).astype(np.int64)
    slopes = _cluster_slopes(gx, gy, order0, bounds, n.astype(np.int64), num_clusters)
    if ((n[[0, -1]] < 2) & (ideal[[0, -1]] >= 2)).any():
        print("Boundary cluster has fewer than 2 points.The chip is unlikely to be segmented correctly.")
    slope = np.nanmedian(slopes)
    # median of y - slope x per cluster: sort the residuals inside their clusters, take the middle one (or two)
    res = gy - slope * gx
    order = np.lexsort((res, lab))
    res = res[order]
    first = np.concatenate([[0], np.cumsum(n)[:-1]]).astype(np.int64)
    cnt = n.astype(np.int64)
    have = cnt > 0
    lo = np.where(have, first + (cnt - 1) // 2, 0)
    hi = np.where(have, first + cnt // 2, 0)
    intercepts = np.where(have, 0.5 * (res[lo] + res[hi]) if len(res) else np.nan, np.nan)
    idx = np.arange(num_clusters)
    g_m, g_b = _line_fit(idx[have], intercepts[have])
    even = g_m * idx + g_b
    with np.errstate(invalid="ignore", divide="ignore"):
        wgt = np.where(have & (ideal != 0), np.minimum(n, ideal) / ideal, 0.0)
    return slope, np.where(wgt > 0, wgt * np.where(have, intercepts, 0.0) + (1 - wgt) * even, even)


def chamber_seed(seed: int, chamber: int, k: int) -> int:
    """RNG stream id of the per-chamber refinement (chamber = i * n_cols + j, k = search channel)."""
    return (seed + 0x51ED270B * (chamber + 1) + 0x2545F491 * (k + 1)) & 0xFFFFFFFFFFFFFFFF


def chamber_seeds(seed: int, m: int, k: int) -> np.ndarray:
    """``chamber_seed`` of chambers 0 .. m - 1 at once (uint64 arithmetic wraps like the mask above)."""
    with np.errstate(over="ignore"):
        return (np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(0x51ED270B) * (np.arange(m, dtype=np.uint64) + np.uint64(1))
                + np.uint64((0x2545F491 * (k + 1)) & 0xFFFFFFFFFFFFFFFF))


def _rounded_centres(y, x):
    """(m, 2) int32 [row, col] = round-half-to-even of the float positions, as Python's ``round`` gives them
    (find.py:316-317)."""
    pos = np.stack([np.rint(np.asarray(y, dtype=np.float64).reshape(-1)), np.rint(np.asarray(x, dtype=np.float64).reshape(-1))], axis=1)
    # a grid fit that degenerated (no circles, clusters without points: NaN slopes) ends here in the reference too:
    # ``round(x[i, j])`` raises for NaN / infinity (find.py:326-327)
    if np.isnan(pos).any():
        raise ValueError("cannot convert float NaN to integer")
    if np.isinf(pos).any() or (np.abs(pos) >= 2.0**30).any():
        raise OverflowError("cannot convert float infinity to integer")
    return pos.astype(np.int32)


def _window_origin(c, L, size):
    """utils.bounding_box for many centres at once: the L-wide window about c, shifted into [0, size)."""
    lo = c.astype(np.int64) - L // 2
    lo = np.where(lo < 0, 0, lo)
    return np.where(lo + L > size, size - L, lo)


class ButtonFinder:
    def __init__(self, row_dist, col_dist, min_button_diameter, max_button_diameter, chamber_diameter, top_chamber,
                 left_chamber, low_edge_quantile, high_edge_quantile, num_iter, min_roundness, cluster_penalty,
                 roi_length, progress_bar, search_timestep, search_channel, interactive):
        if min_button_diameter > max_button_diameter:
            raise ValueError("min_button_diameter must be <= max_button_diameter.")
        if interactive:
            raise NotImplementedError("the napari tuning GUI is outside the MI355X hot path")
        self.row_dist, self.col_dist = row_dist, col_dist
        self.min_button_radius = math.floor(min_button_diameter / 2)
        self.max_button_radius = math.ceil(max_button_diameter / 2)
        self.chamber_radius = round(chamber_diameter / 2)
        self.top_chamber, self.left_chamber = top_chamber, left_chamber
        self.low_edge_quantile, self.high_edge_quantile = low_edge_quantile, high_edge_quantile
        self.num_iter, self.min_roundness, self.cluster_penalty = num_iter, min_roundness, cluster_penalty
        self.roi_length = roi_length if roi_length is not None else round(1.2 * chamber_diameter)
        self.progress_bar = progress_bar
        self.search_timesteps = sorted(utils.to_list(search_timestep))
        self.search_channels = utils.to_list(search_channel)

    # -- find.py:205-306 ---------------------------------------------------------------------------
    def find_centers(self, planes: torch.Tensor, tag: np.ndarray, seeds):
        """planes (n_search, H, W) on the device -> mark_x, mark_y (R, Cc) float64."""
        n_s, h, w = planes.shape
        finder = _finder(1, h, w, self.min_button_radius, self.max_button_radius, self.num_iter, planes.device)
        points = np.empty((0, 2))
        for k in range(n_s):
            res, _ = finder.find(planes[k : k + 1], None, self.low_edge_quantile, self.high_edge_quantile,
                                 self.min_roundness, self.chamber_radius, [seeds[k]])
            new = res[0][0][:, :2].astype(np.float64)
            if len(points) > 0 and len(new) > 0:
                dist = np.linalg.norm(points[np.newaxis] - new[:, np.newaxis], axis=2)
                new = new[np.min(dist, axis=1) > self.chamber_radius]
            points = np.concatenate([points, new])
        x, y = points[:, 1], points[:, 0]
        per_row, per_col = (tag != "").sum(axis=1), (tag != "").sum(axis=0)
        n_rows, n_cols = tag.shape
        if self.top_chamber is None:
            row_labels = hotpath.cluster_1d(y, h, n_rows, self.row_dist, per_row, self.cluster_penalty, planes.device)
        else:
            row_labels = label_clusters(y, self.top_chamber, n_rows, 2 * self.chamber_radius,
                                        self.row_dist - 2 * self.chamber_radius)
        if self.left_chamber is None:
            col_labels = hotpath.cluster_1d(x, w, n_cols, self.col_dist, per_col, self.cluster_penalty, planes.device)
        else:
            col_labels = label_clusters(x, self.left_chamber, n_cols, 2 * self.chamber_radius,
                                        self.col_dist - 2 * self.chamber_radius)
        inside = (row_labels >= 0) & (col_labels >= 0)
        x, y, row_labels, col_labels = x[inside], y[inside], row_labels[inside], col_labels[inside]
        row_slope, row_b = regress_clusters(x, y, row_labels, n_rows, per_row)
        col_slope, col_b = regress_clusters(y, x, col_labels, n_cols, per_col)
        row_b, col_b = np.atleast_1d(row_b), np.atleast_1d(col_b)
        mark_y = (row_slope * col_b[np.newaxis] + row_b[:, np.newaxis]) / (1 - row_slope * col_slope)
        mark_x = mark_y * col_slope + col_b[np.newaxis]
        return mark_x, mark_y

    # -- find.py:308-402 (refinement part) ---------------------------------------------------------
    def refine(self, image_t: torch.Tensor, x, y, tag, search_idx, seed):
        """image_t (C, H, W); grid estimate x, y (R, Cc) -> refined x, y and button radii (R, Cc)."""
        n_c, h, w = image_t.shape
        n_rows, n_cols = tag.shape
        m, L = n_rows * n_cols, self.roi_length
        x, y = x.copy(), y.copy()
        radius = np.full((n_rows, n_cols), self.max_button_radius, dtype=np.int64)
        centers = _rounded_centres(y, x)
        tiles = hotpath.roi_gather_reduce(image_t.view(1, n_c, 1, h, w), [centers], L, None, want_masks=False,
                                          want_sums=False)["roi"]  # (m, C, 1, L, L)
        per_iter = self.num_iter // m
        best_score = np.full(m, -np.inf)
        best = np.full((m, 3), -1, dtype=np.int64)
        if per_iter > 0:
            finder = _finder(m, L, L, self.min_button_radius, self.max_button_radius, per_iter, image_t.device)
            hi_q = 1 - np.pi * self.min_button_radius / L**2
            for k, ch in enumerate(search_idx):
                # only every chamber's best circle is needed (the lists come out sorted by score: [0] is np.argmax):
                # the first row of the device tables, not 784 host lists
                counts, (d_out, d_scores, _) = finder.find(tiles[:, ch, 0], None, self.low_edge_quantile, hi_q,
                                                           self.min_roundness, 0, chamber_seeds(seed, m, k),
                                                           host_results=False)
                top_circle = d_out[:, 0].cpu().numpy().astype(np.int64)
                top_score = d_scores[:, 0].cpu().numpy().astype(np.float64)
                better = (np.asarray(counts) > 0) & (top_score > best_score)
                best[better], best_score[better] = top_circle[better], top_score[better]
        # window origins of all chambers at once (utils.bounding_box: shift the L x L window into the image)
        top, left = _window_origin(centers[:, 0], L, h), _window_origin(centers[:, 1], L, w)
        found = ((tag != "").reshape(-1)) & (best[:, 2] >= 0)
        ys, xs, rs = y.reshape(-1), x.reshape(-1), radius.reshape(-1)  # views of the (contiguous) copies
        ys[found], xs[found] = (best[:, 0] + top)[found], (best[:, 1] + left)[found]
        rs[found] = best[found, 2]
        return x, y, radius

    def __call__(self, assay):
        """find.py:55-203."""
        image = _image_tensor(assay)
        n_c, n_t, h, w = image.shape
        if not self.search_channels:
            self.search_channels = (assay.coords["channel"].values.tolist() if "channel" in assay.coords
                                    else list(range(n_c)))
        search_idx = [_channel_index(assay, c) for c in self.search_channels]
        tag = assay.coords["tag"].values
        n_rows, n_cols = tag.shape
        m, L = n_rows * n_cols, self.roi_length
        img_t = image.permute(1, 0, 2, 3).contiguous()  # (T, C, H, W): one assay-like block per timestep
        x = np.empty((n_rows, n_cols, n_t))
        y = np.empty((n_rows, n_cols, n_t))
        radius = np.empty((n_rows, n_cols, n_t), dtype=np.int64)
        valid = assay.coords["valid"].values.copy()
        for t in self.search_timesteps:
            planes = torch.stack([img_t[t, c] for c in search_idx])
            gx, gy = self.find_centers(planes, tag, [utils.next_seed() for _ in search_idx])
            x[..., t], y[..., t], radius[..., t] = self.refine(img_t[t], gx, gy, tag, search_idx, utils.next_seed())
        for t in range(n_t):
            if t in self.search_timesteps:
                continue
            src = self.search_timesteps[0] if t < self.search_timesteps[0] else t - 1
            x[..., t], y[..., t], radius[..., t], valid[..., t] = x[..., src], y[..., src], radius[..., src], valid[..., src]
        # final windows, masks and sums for every (timestep, chamber)
        centers = [_rounded_centres(y[..., t], x[..., t]) for t in range(n_t)]
        out = hotpath.roi_gather_reduce(img_t.view(n_t, n_c, 1, h, w), centers, L, None, want_masks=False,
                                        want_sums=False)
        roi = out["roi"].view(n_t, m, n_c, L, L).permute(1, 2, 0, 3, 4).contiguous()  # (M, C, T, L, L)
        every = np.concatenate(centers)  # (T * m, 2): position of every centre inside its own window
        rel = np.stack([every[:, 0] - _window_origin(every[:, 0], L, h), every[:, 1] - _window_origin(every[:, 1], L, w)],
                       axis=1).astype(np.int32)
        radii_tm = np.ascontiguousarray(radius.reshape(m, n_t).T).reshape(-1)
        fg, bg = hotpath.button_masks(rel, radii_tm, L, self.chamber_radius, self.max_button_radius, image.device)
        fg = fg.view(n_t, m, L, L).permute(1, 0, 2, 3).contiguous()  # (M, T, L, L)
        bg = bg.view(n_t, m, L, L).permute(1, 0, 2, 3).contiguous()
        grid = (n_rows, n_cols)
        assay["roi"] = DataArray(roi.view(grid + (n_c, n_t, L, L)),
                                 ("mark_row", "mark_col", "channel", "time", "roi_y", "roi_x"))
        assay = assay.assign_coords(
            fg=(("mark_row", "mark_col", "time", "roi_y", "roi_x"), fg.view(grid + (n_t, L, L)).bool()),
            bg=(("mark_row", "mark_col", "time", "roi_y", "roi_x"), bg.view(grid + (n_t, L, L)).bool()),
            x=(("mark_row", "mark_col", "time"), x), y=(("mark_row", "mark_col", "time"), y),
            valid=(("mark_row", "mark_col", "time"), valid),
        )
        assay._cache["radius"] = radius
        assay = assay.stack_mark()
        # rechunked along the markers after stacking, as the reference does (find.py:182-201; bounded by the ROW count)
        per = {"mark": utils.roi_mark_chunk(n_rows, n_c, n_t, L)}
        assay["roi"] = assay.data_vars["roi"].chunk(per)
        for k in ("fg", "bg"):
            assay.coords[k] = assay.coords[k].chunk(per)
        return assay

    @registry.components.register("find_buttons")
    def make(row_dist, col_dist, min_button_diameter, max_button_diameter, chamber_diameter, top_chamber, left_chamber,
             low_edge_quantile, high_edge_quantile, num_iter, min_roundness, cluster_penalty, roi_length, progress_bar,
             search_timestep, search_channel, interactive):
        return ButtonFinder(row_dist=row_dist, col_dist=col_dist, min_button_diameter=min_button_diameter,
                            max_button_diameter=max_button_diameter, chamber_diameter=chamber_diameter,
                            top_chamber=top_chamber, left_chamber=left_chamber, low_edge_quantile=low_edge_quantile,
                            high_edge_quantile=high_edge_quantile, num_iter=num_iter, min_roundness=min_roundness,
                            cluster_penalty=cluster_penalty, roi_length=roi_length, progress_bar=progress_bar,
                            search_timestep=search_timestep, search_channel=search_channel, interactive=interactive)
