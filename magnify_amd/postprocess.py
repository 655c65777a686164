"""drop and restore_format (reference: src/magnify/postprocess.py:6-49)."""
from __future__ import annotations


from . import registry
from .xr_lite import DataArray, Dataset

STANDARD_DIMS = ["channel", "time", "tile_row", "tile_col", "tile_y", "tile_x"]


@registry.component("drop")
def drop(xp, roi_only: bool = False, drop_tiles: bool = True):
    if roi_only:
        return xp.roi.assign_attrs(xp.attrs)
    elif drop_tiles:
        return xp.drop_vars(["tile", "tile_row", "tile_col"], errors="ignore")
    return xp


def _unstack_time(v: DataArray, names, sizes):
    if "time" not in v.dims:
        return v
    ax = v.dims.index("time")
    data = v.data
    shape = tuple(v.shape[:ax]) + tuple(sizes) + tuple(v.shape[ax + 1:])
    dims = v.dims[:ax] + tuple(names) + v.dims[ax + 1:]
    return DataArray(data.reshape(shape), dims, None, v.name, v.attrs)


@registry.component("restore_format")
def restore_format(xp):
    original = list(xp.attrs["__original_tile_dims__"])
    stacked = xp.attrs.get("__mg_stacked_time__")
    if isinstance(xp, Dataset):
        xp = xp.unstack()
        variables = {**{k: ("coord", v) for k, v in xp.coords.items()},
                     **{k: ("var", v) for k, v in xp.data_vars.items()}}
    else:
        variables = {"__self__": ("var", xp)}
    out = {}
    for name, (kind, v) in variables.items():
        if stacked:
            v = _unstack_time(v, *stacked)
            if "__time__" in v.dims:
                v = v.rename({"__time__": "time"})
        # remove dimensions that standardize_format added
        added = [dim for dim in STANDARD_DIMS if dim not in original and dim in v.dims]
        if added:
            v = v.squeeze(added)
        # restore the original relative order of the original dimensions
        orig = [d for d in original if d in v.dims]
        if orig:
            dims = list(v.dims)
            idxs = [i for i, d in enumerate(dims) if d in orig]
            start, end = idxs[0], idxs[-1] + 1
            v = v.transpose(*(dims[:start] + orig + [d for d in dims[start:end] if d not in orig] + dims[end:]))
        out[name] = (kind, v)
    attrs = {k: a for k, a in xp.attrs.items() if k not in ("__original_tile_dims__", "__mg_stacked_time__")}
    if not isinstance(xp, Dataset):
        res = out["__self__"][1]
        res.attrs = attrs
        return res
    res = Dataset(attrs=attrs)
    res._cache.update(xp._cache)
    for name, (kind, v) in out.items():
        v.name = name
        (res.coords if kind == "coord" else res.data_vars)[name] = DataArray(v.raw, v.dims, None, name, v.attrs)
    return res
