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


def _sums_counts(xp, mask):
    """(sums (M, C, T) float64, counts (M, T) int64) of roi under a mask, by the masked-sum kernel (mg_masked_sums):
    one call when the mask is the same at every timepoint (beads: geometry replicated over time, find.py:585-586),
    one per timepoint otherwise (chips searched at several timesteps)."""
    roi = xp.data_vars["roi"].transpose("mark", "channel", "time", "roi_y", "roi_x").data
    if not isinstance(roi, torch.Tensor):
        roi = torch.from_numpy(np.ascontiguousarray(roi))
    roi = roi.cuda()
    m = _masks(xp, mask)
    m = m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)
    n_m, n_c, n_t = roi.shape[:3]
    if n_t == 1 or m.stride(1) == 0:
        one = m[:, 0].contiguous()
        sums, cnt = hotpath.masked_sums(roi.contiguous(), one, one)
        return sums[..., 0], cnt[:, :1].to(torch.int64).expand(n_m, n_t)
    sums = torch.empty((n_m, n_c, n_t), dtype=torch.float64, device=roi.device)
    cnt = torch.empty((n_m, n_t), dtype=torch.int64, device=roi.device)
    for t in range(n_t):
        one = m[:, t].contiguous()
        s_t, c_t = hotpath.masked_sums(roi[:, :, t:t + 1].contiguous(), one, one)
        sums[:, :, t] = s_t[:, :, 0, 0]
        cnt[:, t] = c_t[:, 0]
    return sums, cnt


def counts(xp, mask="fg"):
    """xp.fg.sum(dim=["roi_x", "roi_y"])  (README.md:21) -> (mark, time) int64."""
    cached = xp._cache.get("roi_counts")
    n_t = xp.sizes.get("time", 1)
    if cached is not None and mask in ("fg", "bg") and "mark" in xp.sizes and cached.shape[0] == xp.sizes["mark"]:
        return DataArray(cached[:, 0 if mask == "fg" else 1].to(torch.int64)[:, None].expand(-1, n_t), ("mark", "time"))
    return DataArray(_sums_counts(xp, mask)[1], ("mark", "time"))


def masked_sum(xp, mask="fg"):
    """roi.where(mask).sum(dim=[roi_y, roi_x]) -> (mark, channel, time) float64 (exact for integers)."""
    cached = xp._cache.get("roi_sums")
    if cached is not None and mask in ("fg", "bg"):
        return DataArray(cached[..., 0 if mask == "fg" else 1], ("mark", "channel", "time"))
    return DataArray(_sums_counts(xp, mask)[0], ("mark", "channel", "time"))


def masked_mean(xp, mask="fg"):
    """roi.where(mask).mean(dim=[roi_y, roi_x]) (README.md:22): sum / count, NaN for an empty mask."""
    s = masked_sum(xp, mask).data
    n = counts(xp, mask).data.to(torch.float64)[:, None]
    return DataArray(s / n, ("mark", "channel", "time"))


def masked_median(xp, mask="bg", time=None):
    """roi.where(mask).median(dim=[roi_y, roi_x]) with numpy nanmedian semantics -> (mark, channel, time) float64.
    Any roi dtype the hot path carries (uint8, uint16, float32, float64); the masks may differ between timepoints (a
    chip searched at several timesteps, find.py:119-140).  ``time=k``: of timepoint ``k`` only, selected BEFORE the
    reduction as ``assay.isel(time=0)`` does in filter.py:20-22, 69-75 and identify.py:76 -- the other timepoints are
    neither read nor reduced; the result keeps a time axis of length 1."""
    roi = xp.data_vars["roi"].transpose("mark", "channel", "time", "roi_y", "roi_x").data
    if not isinstance(roi, torch.Tensor):
        roi = np.ascontiguousarray(roi)
        if roi.dtype.byteorder not in ("=", "|"):
            roi = roi.astype(roi.dtype.newbyteorder("="))
        roi = torch.from_numpy(roi)
    hotpath.nat.dtype_code(roi.dtype)  # TypeError for anything but uint8 / uint16 / float32 / float64
    m = _masks(xp, mask)
    if time is not None:
        roi = roi[:, :, time:time + 1] if time != -1 else roi[:, :, -1:]
        m = m[:, time:time + 1] if m.shape[1] != 1 else m
    return DataArray(hotpath.masked_median(roi.cuda(), m.cuda()), ("mark", "channel", "time"))


def fg_mean_minus_bg_median(xp, time=None):
    """identify.py:76-80 (``time=0``: ``assay.roi.isel(time=0)`` first, as the reference does)."""
    mean = masked_mean(xp, "fg").data
    if time is not None:
        mean = mean[:, :, time:time + 1]
    return DataArray(mean - masked_median(xp, "bg", time=time).data, ("mark", "channel", "time"))
