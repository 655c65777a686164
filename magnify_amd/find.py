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
        for channel in self.search_channels:
            ch = _channel_index(assay, channel)
            res, _ = finder.find(image[ch, 0:1], _plane_minmax(assay, image, (ch, 0)), self.low_edge_quantile,
                                 self.high_edge_quantile, self.min_roundness, self.min_bead_radius,
                                 [utils.next_seed()])
            b = dedup_against(beads, res[0][0], 2 * self.min_bead_radius)  # find.py:490-500
            beads = np.concatenate([beads, b])
        m, L = len(beads), self.roi_length
        labels = hotpath.circle_labels([beads], h, w, device=image.device) if m else None
        out = hotpath.roi_gather_reduce(image[None], [beads], L, labels)
        fg = out["fg"].bool()[:, None].expand(m, n_t, L, L)  # geometry replicated over time (find.py:585-586)
        bg = out["bg"].bool()[:, None].expand(m, n_t, L, L)
        assay["roi"] = DataArray(out["roi"], ("mark", "channel", "time", "roi_y", "roi_x"))
        xy = beads.astype(np.float64)
        assay = assay.assign_coords(
            fg=(("mark", "time", "roi_y", "roi_x"), fg),
            bg=(("mark", "time", "roi_y", "roi_x"), bg),
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
