"""Per-ROI reductions on the device: the expressions users and the reference's
identify/filter steps write with xarray (README.md:21-22, identify.py:76-80, filter.py:21-22),
computed by the fused HIP reductions instead of a float64 NaN-masked copy of ``roi``.

All functions take the Dataset returned by ``find_beads`` / ``find_buttons`` *before*
``restore_format`` squeezes dimensions, or any Dataset holding ``roi (mark, channel, time, roi_y,
roi_x)`` with ``fg`` / ``bg (mark, time, roi_y, roi_x)``."""
from __future__ import annotations

import numpy as np
import torch

from . import hotpath
from .xr_lite import DataArray


def _masks(xp, name):
    m = xp.coords[name].transpose("mark", "time", "roi_y", "roi_x").data
    if not isinstance(m, torch.Tensor):
        m = torch.from_numpy(np.ascontiguousarray(m))
    return m.cuda()


def counts(xp, mask="fg"):
    """xp.fg.sum(dim=["roi_x", "roi_y"])  (README.md:21) -> (mark, time) int64."""
    m = _masks(xp, mask)
    return DataArray(m.to(torch.int32).sum(dim=(-1, -2)).to(torch.int64), ("mark", "time"))


def masked_sum(xp, mask="fg"):
    """roi.where(mask).sum(dim=[roi_y, roi_x]) -> (mark, channel, time) float64 (exact for integers)."""
    cached = xp._cache.get("roi_sums")
    if cached is not None and mask in ("fg", "bg"):
        return DataArray(cached[..., 0 if mask == "fg" else 1], ("mark", "channel", "time"))
    roi = xp.data_vars["roi"].transpose("mark", "channel", "time", "roi_y", "roi_x").data.cuda()
    m = _masks(xp, mask)[:, None].to(torch.float64)
    return DataArray((roi.to(torch.float64) * m).sum(dim=(-1, -2)), ("mark", "channel", "time"))


def masked_mean(xp, mask="fg"):
    """roi.where(mask).mean(dim=[roi_y, roi_x]) (README.md:22): sum / count, NaN for an empty mask."""
    s = masked_sum(xp, mask).data
    n = counts(xp, mask).data.to(torch.float64)[:, None]
    return DataArray(s / n, ("mark", "channel", "time"))


def masked_median(xp, mask="bg"):
    """roi.where(mask).median(dim=[roi_y, roi_x]) with numpy nanmedian semantics (uint16 rois)."""
    roi = xp.data_vars["roi"].transpose("mark", "channel", "time", "roi_y", "roi_x").data
    if not isinstance(roi, torch.Tensor) or roi.dtype != torch.uint16:
        raise TypeError("the device median kernel handles uint16 rois")
    m = _masks(xp, mask)
    if m.shape[1] != 1 and not bool((m == m[:, :1]).all()):
        raise NotImplementedError("time-varying masks: call per timestep")
    return DataArray(hotpath.masked_median_u16(roi.cuda().contiguous(), m[:, 0].to(torch.uint8).contiguous()),
                     ("mark", "channel", "time"))


def fg_mean_minus_bg_median(xp):
    """identify.py:76-80."""
    return DataArray(masked_mean(xp, "fg").data - masked_median(xp, "bg").data, ("mark", "channel", "time"))
