"""standardize_format and flatfield_correct (reference: src/magnify/preprocess.py:11-41, 62-88),
plus the trivial view components that keep their registry names (rotate is a no-op in the
reference as well, preprocess.py:54-59)."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import _native as nat
from . import hotpath, registry, xr_lite
from .xr_lite import DataArray, Dataset

DESIRED_ORDER = ["channel", "time", "tile_row", "tile_col", "tile_y", "tile_x"]
SUPPORTED = ("uint8", "uint16", "float32", "float64")


def to_device(data):
    """NumPy / torch array -> contiguous device tensor of a supported dtype."""
    if isinstance(data, torch.Tensor):
        t = data
    else:
        arr = np.asarray(data)
        if arr.dtype.name not in SUPPORTED:
            if arr.dtype.kind in "iub":
                arr = arr.astype(np.float64)  # exact below 2**53
            else:
                raise TypeError(f"unsupported image dtype {arr.dtype}")
        t = torch.from_numpy(np.ascontiguousarray(arr))
    if str(t.dtype).replace("torch.", "") not in SUPPORTED:
        raise TypeError(f"unsupported image dtype {t.dtype}")
    hotpath.require_gpu()
    return t.cuda().contiguous()


@registry.component("standardize_format")
def standardize_format(xp):
    xp = xr_lite.from_any(xp)
    if isinstance(xp, DataArray):
        ds = Dataset(attrs=xp.attrs)
        for k, c in xp.coords.items():
            ds.coords[k] = c
        ds["tile"] = xp
        xp = ds
    for old in ["x", "y", "row", "col"]:
        if old in xp.tile.dims:
            xp = xp.rename({old: "tile_" + old})
    xp.attrs["__original_tile_dims__"] = list(xp.tile.dims)
    tile = xp.data_vars["tile"]
    extra = [d for d in tile.dims if d not in DESIRED_ORDER]
    if extra:
        # stack every additional dimension (and an existing time axis) into one time axis
        # (preprocess.py:24-32)
        if "time" in tile.dims:
            xp = xp.rename(time="__time__")
            tile = xp.data_vars["tile"]
            extra.append("__time__")
        tile = tile.transpose(*extra, ...)
        sizes = [tile.sizes[d] for d in extra]
        data = tile.data.reshape((int(np.prod(sizes)),) + tuple(tile.shape[len(extra):]))
        xp.attrs["__mg_stacked_time__"] = (list(extra), sizes)
        for d in extra:
            xp.coords.pop(d, None)
        xp.data_vars["tile"] = DataArray(data, ("time",) + tile.dims[len(extra):], name="tile")
    tile = xp.data_vars["tile"]
    for dim in DESIRED_ORDER:
        if dim not in tile.dims:
            tile = tile.expand_dims(dim)
    xp.data_vars["tile"] = DataArray(tile.transpose(*DESIRED_ORDER).raw, DESIRED_ORDER, name="tile")
    return xp


@registry.component("rename_labels")
def rename_labels(xp, **coords):
    for name, new in coords.items():
        if isinstance(new, dict):
            new = [new.get(v, v) for v in xp[name].values.tolist()]
        xp = xp.assign_coords({name: new})
    return xp


@registry.component("rotate")
def rotate(xp, rotation=0):
    return xp  # no-op in the reference too (preprocess.py:54-59)


class LazyFlatfield:
    """The flat-field correction as a pending operation on the tile stack -- the counterpart of the
    reference's lazy dask expression (preprocess.py:83-87), which is only executed when ``stitch``
    caches the image (stitch.py:45).  ``stitch`` fuses it with the crop/concat in one kernel;
    any other access materialises it with the same kernel (overlap 0 per tile)."""

    def __init__(self, tiles: torch.Tensor, flatfield, darkfield):
        self.tiles, self.flatfield, self.darkfield = tiles, flatfield, darkfield
        self.shape, self.dtype = tuple(tiles.shape), tiles.dtype
        # pass 1: the two global maxima (not needed when the correction is the identity: integer pixels, flat 1, dark 0)
        self.max2 = (None if hotpath.flatfield_is_identity(tiles.dtype, flatfield, darkfield)
                     else hotpath.flatfield_max(tiles, flatfield, darkfield))

    def materialize(self):
        c, t, nr, nc, ty, tx = self.shape
        flat = self.tiles.reshape(c * t * nr * nc, 1, 1, 1, ty, tx)
        out, _ = hotpath.flatfield_stitch(flat, 0, self.flatfield, self.darkfield, max2=self.max2, want_minmax=False)
        return out.reshape(self.shape)


def _field(value):
    """Flat / dark field operand: a scalar, an image, or -- as in the reference (preprocess.py:64-81, read with
    tifffile there) -- the path of a single-page TIFF holding the image.  The reference's zarr-directory branch
    (preprocess.py:66-73) calls a non-existent ``xr.dataset.open_zarr`` and passes a DataArray as a path: it
    cannot run there either and raises here."""
    if isinstance(value, (str, os.PathLike)):
        import pathlib

        path = pathlib.Path(value).expanduser()
        if path.is_dir():
            raise NotImplementedError("flat-field zarr directories: the reference's own branch for them "
                                      "(preprocess.py:66-73) is broken; pass a TIFF file or an array")
        if not path.exists():
            raise FileNotFoundError(str(path))
        from . import tiff

        with tiff.TiffFile(str(path)) as tif:  # (tifffile.imread in the reference: every page of the file)
            pages = [tif.asarray(i) for i in range(len(tif))]
        return pages[0] if len(pages) == 1 else np.stack(pages)
    if isinstance(value, (DataArray,)):
        value = value.values
    return value


@registry.component("flatfield_correct")
def flatfield_correct(xp, flatfield=1.0, darkfield=0.0):
    tiles = to_device(xp.data_vars["tile"].data)
    xp.data_vars["tile"] = DataArray(LazyFlatfield(tiles, _field(flatfield), _field(darkfield)),
                                     xp.data_vars["tile"].dims, name="tile")
    return xp


@registry.component("horizontal_flip")
def horizontal_flip(xp):
    name, dim = ("image", "im_x") if "image" in xp else ("tile", "tile_x")
    v = xp.data_vars[name]
    ax = v.dims.index(dim)
    data = v.data
    data = torch.flip(data, (ax,)) if isinstance(data, torch.Tensor) else np.flip(data, ax)
    xp.data_vars[name] = DataArray(data, v.dims, name=name)
    return xp


@registry.component("vertical_flip")
def vertical_flip(xp):
    name, dim = ("image", "im_y") if "image" in xp else ("tile", "tile_y")
    v = xp.data_vars[name]
    ax = v.dims.index(dim)
    data = v.data
    data = torch.flip(data, (ax,)) if isinstance(data, torch.Tensor) else np.flip(data, ax)
    xp.data_vars[name] = DataArray(data, v.dims, name=name)
    return xp


@registry.component("circle_mask")
def circle_mask(xp, center, diameter, mask_inner=False):
    """preprocess.py:136-153: keep (or, with ``mask_inner``, blank) the filled ``cv.circle`` of the
    given diameter about ``center = (row, col)`` in every plane of ``image`` (or ``tile``)."""
    name = "image" if "image" in xp else "tile"
    v = xp.data_vars[name]
    h, w = v.shape[-2:]
    radius = int(diameter) // 2
    half = nat.cv_disk_halfwidths(radius)  # max |dx| for |dy| = 0..radius (cv.circle, thickness=-1)
    mask = np.zeros((h, w), dtype=bool)
    cy, cx = int(center[0]), int(center[1])
    for ady in range(radius + 1):
        if half[ady] < 0:
            continue
        for y in {cy - ady, cy + ady}:
            if 0 <= y < h:
                x0, x1 = max(cx - int(half[ady]), 0), min(cx + int(half[ady]), w - 1)
                if x0 <= x1:
                    mask[y, x0 : x1 + 1] = True
    if mask_inner:
        mask = ~mask
    data = v.data
    if isinstance(data, torch.Tensor):
        keep = torch.from_numpy(mask).to(data.device)
        data = torch.where(keep, data, torch.zeros((), dtype=data.dtype, device=data.device))
    else:
        data = data * mask.astype(data.dtype)
    xp.data_vars[name] = DataArray(data, v.dims, name=name)
    return xp


@registry.component("basic_correct")
def basic_correct(xp):
    """preprocess.py:91-115 fits a BaSiC illumination model with the third-party ``basicpy``; not part of
    this build (use ``flatfield_correct`` with measured flat / dark images)."""
    raise NotImplementedError("basic_correct needs the basicpy package; use flatfield_correct")
