"""``mg.save`` / ``mg.load`` (reference: src/magnify/file.py:6-17).

The reference writes with ``xarray.Dataset.to_netcdf``.  Neither xarray nor a NetCDF-4 library is
available to this build, so the file is NetCDF-3 (64-bit offset) written through
``scipy.io.netcdf_file`` -- the format xarray's own ``scipy`` engine produces -- using xarray's
encoding conventions, so that ``xr.open_dataset`` (and therefore the reference's ``mg.load``)
decodes it: unsigned integers as the signed type of the same width with ``_Unsigned = "true"``,
booleans as int8 with ``dtype = "bool"``, strings as char arrays over a ``string<N>`` dimension with
``_Encoding = "utf-8"``, int64 narrowed to int32 when it fits, non-index coordinates listed in the
``coordinates`` attribute.  NetCDF-3 limits a variable to 4 GiB: a dataset with a larger variable (C4's
``roi`` is 9.7 GB) is written as PARTS ``<file>.part000``, ``<file>.part001`` ... -- contiguous blocks of the
marker axis, every part a complete NetCDF file of at most ``shard_bytes`` per variable -- and ``load`` puts
them together again (ROI pixels stay sharded by GPU anyway, SURVEY 8e: a rank saves its own shard).

As in the reference, a chip dataset is unstacked (mark -> mark_row, mark_col) before saving and
restacked on load.
"""
from __future__ import annotations

import numpy as np

from .xr_lite import DataArray, Dataset

_LIMIT = (1 << 32) - 4


def _encode(name, arr: np.ndarray):
    """-> (array NetCDF-3 can hold, extra dims, attributes)."""
    attrs = {}
    extra = ()
    if arr.dtype == np.bool_:
        arr, attrs["dtype"] = arr.astype(np.int8), "bool"
    elif arr.dtype.kind == "u":
        signed = {1: np.int8, 2: np.int16, 4: np.int32}.get(arr.dtype.itemsize)
        if signed is None:
            raise TypeError(f"{name}: uint64 cannot be stored in NetCDF-3")
        arr, attrs["_Unsigned"] = arr.view(signed), "true"
    elif arr.dtype == np.int64:
        if arr.size and (arr.min() < -(2**31) or arr.max() >= 2**31):
            raise ValueError(f"{name}: int64 values do not fit the int32 of NetCDF-3")
        arr = arr.astype(np.int32)
    elif arr.dtype.kind in "US":
        raw = np.char.encode(arr.astype(str), "utf-8") if arr.dtype.kind == "U" else arr
        width = max(raw.dtype.itemsize, 1)
        raw = raw.astype(f"S{width}")
        arr = raw.view("S1").reshape(raw.shape + (width,))
        extra = (f"string{width}",)
        attrs["_Encoding"] = "utf-8"
    elif arr.dtype.kind == "f" and arr.dtype.itemsize not in (4, 8):
        arr = arr.astype(np.float32)
    elif arr.dtype.kind not in "if":
        raise TypeError(f"{name}: dtype {arr.dtype} cannot be stored in NetCDF-3")
    if arr.nbytes > _LIMIT:
        raise ValueError(f"{name}: {arr.nbytes} bytes exceed the 4 GiB variable limit of NetCDF-3; save per shard")
    return np.ascontiguousarray(arr), extra, attrs


def _decode(var, attrs):
    arr = np.array(var.data)
    dims = tuple(var.dimensions)
    if arr.dtype.byteorder == ">":
        arr = arr.astype(arr.dtype.newbyteorder("="))
    if attrs.pop("_Unsigned", None) in ("true", b"true"):
        arr = arr.view({1: np.uint8, 2: np.uint16, 4: np.uint32}[arr.dtype.itemsize])
    if attrs.pop("dtype", None) in ("bool", b"bool"):
        arr = arr.astype(bool)
    if arr.dtype.kind == "S" and dims and dims[-1].startswith("string"):
        attrs.pop("_Encoding", None)
        width = arr.shape[-1]
        arr = np.char.decode(np.ascontiguousarray(arr).view(f"S{width}").reshape(arr.shape[:-1]), "utf-8")
        dims = dims[:-1]
    return arr, dims


_SHARD = 1 << 31  # default size of the largest variable of a part


def _nbytes(v):
    """Bytes of a variable as it will be stored -- from its shape and dtype (a device-resident array stays where it is)."""
    dtype = np.dtype(v.dtype)
    return int(np.prod(v.shape, dtype=np.int64)) * (1 if dtype == np.bool_ else dtype.itemsize)


def _take_block(ds: Dataset, dim, lo, hi, first):
    """The block [lo, hi) of ``ds`` along ``dim``; variables without that dimension only in the first part."""
    out = Dataset(attrs=dict(ds.attrs))

    def cut(v):
        if dim not in v.dims:
            return v if first else None
        idx = [slice(None)] * len(v.dims)
        idx[v.dims.index(dim)] = slice(lo, hi)
        # (sliced where the array lives: a device-resident variable comes to the host part by part, when it is written)
        return DataArray(v.data[tuple(idx)], v.dims, None, v.name, dict(v.attrs or {}))

    for k, c in ds.coords.items():
        piece = cut(c)
        if piece is not None:
            out.coords[k] = piece
    for k, v in ds.data_vars.items():
        piece = cut(v)
        if piece is not None:
            out[k] = piece
    return out


def save(file, xp, shard_bytes=None):
    """file.py:6-8.  ``shard_bytes``: write parts whose largest variable stays below it (default: parts only when
    a variable exceeds NetCDF-3's 4 GiB, then 2 GiB each).  Everything is written under temporary names first and
    renamed when complete; what an earlier save left under this name (the whole file, or parts -- possibly more of
    them) is removed only AFTER the new data is in place: a crash or a full disk during the write leaves the old
    data untouched."""
    ds = xp.unstack() if isinstance(xp, Dataset) else Dataset({xp.name or "data": xp})
    sizes = {k: _nbytes(v) for k, v in list(ds.data_vars.items()) + list(ds.coords.items())}
    biggest = max(sizes.values(), default=0)
    if shard_bytes is None and biggest <= _LIMIT:
        return _commit(file, [(str(file), ds)])
    limit = int(shard_bytes or _SHARD)
    dim = next((d for d in ("mark", "mark_row", "time") if d in ds.sizes and ds.sizes[d] > 1), None)
    if dim is None:
        raise ValueError("a variable exceeds the part size and there is no marker / time axis to split along")
    # rows of `dim` per part so that the largest variable carrying it fits
    per_row = max((sizes[k] // max(ds.sizes[dim], 1) for k, v in list(ds.data_vars.items()) + list(ds.coords.items())
                   if dim in v.dims), default=1)
    rows = max(1, limit // max(per_row, 1))
    # parts end at chunk boundaries of the variables' chunk policy (find.py:506-531), when that leaves at least a chunk
    policy = [v.attrs["__mg_chunks__"][dim] for v in list(ds.data_vars.values()) + list(ds.coords.values())
              if dim in v.attrs.get("__mg_chunks__", {})]
    if policy and rows >= max(policy) > 0:
        rows -= rows % max(policy)
    n = ds.sizes[dim]
    bounds = list(range(0, n, rows)) + [n]

    def parts():
        for k, (lo, hi) in enumerate(zip(bounds[:-1], bounds[1:])):
            part = _take_block(ds, dim, lo, hi, k == 0)
            part.attrs.update(mg_part=k, mg_parts=len(bounds) - 1, mg_split_dim=dim, mg_split_lo=lo)
            yield f"{file}.part{k:03d}", part

    _commit(file, parts())


def _stale(file):
    import glob
    import os

    return [old for old in [str(file)] + glob.glob(glob.escape(str(file)) + ".part[0-9][0-9][0-9]") if os.path.isfile(old)]


def _commit(file, named):
    """Write every (final name, dataset) of ``named`` under a temporary name (one at a time: a part comes to the host
    when it is written), then rename them all, then remove what an earlier save left that was not replaced."""
    import os
    import uuid

    tag = f".tmp{os.getpid()}-{uuid.uuid4().hex[:8]}"
    written = []
    try:
        for final, ds in named:
            _write_nc(final + tag, ds)
            written.append(final)
        for final in written:
            os.replace(final + tag, final)
        for old in _stale(file):
            if old not in written:
                os.remove(old)
    finally:
        for final in written:
            if os.path.exists(final + tag):
                os.remove(final + tag)


def _write_nc(file, ds):
    from scipy.io import netcdf_file

    coord_names = [k for k, c in ds.coords.items() if not (c.dims == (k,))]
    with netcdf_file(str(file), "w", version=2) as nc:
        def dim(name, size):
            if name not in nc.dimensions:
                nc.createDimension(name, int(size))

        def put(name, v, is_data_var):
            arr, extra, attrs = _encode(name, np.asarray(v.values))
            dims = tuple(v.dims) + extra
            for d, n in zip(dims, arr.shape):
                dim(d, n)
            var = nc.createVariable(name, "c" if arr.dtype.kind == "S" else arr.dtype, dims)
            if arr.ndim:
                var[...] = arr
            else:
                var.assignValue(arr)
            for k, a in dict(v.attrs or {}, **attrs).items():
                if isinstance(a, (str, bytes, int, float, np.integer, np.floating)):
                    setattr(var, k, a)
            if is_data_var:
                mine = [c for c in coord_names if set(ds.coords[c].dims) <= set(v.dims)]
                if mine:
                    var.coordinates = " ".join(mine)

        for k, v in ds.data_vars.items():
            put(k, v, True)
        for k, c in ds.coords.items():
            put(k, c, False)
        if coord_names:
            nc.coordinates = " ".join(coord_names)
        for k, a in ds.attrs.items():
            if isinstance(a, (str, bytes, int, float, np.integer, np.floating)):
                setattr(nc, k, a)
            elif isinstance(a, (list, tuple)) and all(isinstance(x, str) for x in a):
                setattr(nc, k, " ".join(a))


def load(file):
    """file.py:11-17.  A dataset that was saved in parts (``<file>.part000`` ...) is put together again along the
    axis it was split on."""
    import glob
    import os

    parts = sorted(glob.glob(glob.escape(str(file)) + ".part[0-9][0-9][0-9]"))
    if parts and os.path.exists(str(file)):
        raise ValueError(f"{file}: both the file and parts of it exist (two different saves); remove one")
    if parts or not os.path.exists(str(file)):
        if not parts:
            raise FileNotFoundError(str(file))
        pieces = [_read(p, restack=False) for p in parts]
        n, dim = int(pieces[0].attrs["mg_parts"]), pieces[0].attrs["mg_split_dim"]
        if len(pieces) != n or [int(p.attrs["mg_part"]) for p in pieces] != list(range(n)):
            raise ValueError(f"{file}: parts missing (found {len(pieces)} of {n})")
        xp = Dataset(attrs={k: v for k, v in pieces[0].attrs.items() if not k.startswith("mg_")})

        def join(name, get):
            first = get(pieces[0])
            if dim not in first.dims:
                return first
            axis = first.dims.index(dim)
            return DataArray(np.concatenate([np.asarray(get(p).values) for p in pieces], axis=axis), first.dims, None,
                             name, dict(first.attrs or {}))

        for k in pieces[0].coords:
            xp.coords[k] = join(k, lambda p, k=k: p.coords[k])
        for k in pieces[0].data_vars:
            xp[k] = join(k, lambda p, k=k: p.data_vars[k])
        if "mark_row" in xp.sizes and "mark_col" in xp.sizes:
            xp = xp.stack_mark()
        return xp
    return _read(file, restack=True)


def _read(file, restack):
    from scipy.io import netcdf_file

    def text(a):
        return a.decode() if isinstance(a, bytes) else a

    with netcdf_file(str(file), "r", mmap=False) as nc:
        listed = set(text(getattr(nc, "coordinates", "")).split())
        decoded = {}
        for name, var in nc.variables.items():
            attrs = {k: text(v) for k, v in var._attributes.items()}
            listed |= set(attrs.pop("coordinates", "").split())
            arr, dims = _decode(var, attrs)
            decoded[name] = DataArray(arr, dims, None, name, attrs)
        global_attrs = {k: text(v) for k, v in nc._attributes.items() if k != "coordinates"}
    coords = {k: v for k, v in decoded.items() if k in listed or v.dims == (k,)}
    data_vars = {k: v for k, v in decoded.items() if k not in coords}
    xp = Dataset(data_vars, coords=coords, attrs=global_attrs)
    if restack and "mark_row" in xp.sizes and "mark_col" in xp.sizes:
        xp = xp.stack_mark()
    return xp
