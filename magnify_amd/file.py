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

import struct as _struct
import sys as _sys

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
        # (the byte order stays: a sink's block may already be big-endian, swapped on the device)
        arr, attrs["_Unsigned"] = arr.view(np.dtype(signed).newbyteorder(arr.dtype.byteorder)), "true"
    elif arr.dtype.kind == "i" and arr.dtype.itemsize == 8:
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


def save(file, xp, shard_bytes=None, threads=None):
    """file.py:6-8.  ``shard_bytes``: write parts whose largest variable stays below it (default: parts only when
    a variable exceeds NetCDF-3's 4 GiB, then 2 GiB each).  Everything is written under temporary names first and
    renamed when complete; what an earlier save left under this name (the whole file, or parts -- possibly more of
    them) is removed only AFTER the new data is in place: a crash or a full disk during the write leaves the old
    data untouched.  ``threads``: writer threads for one file's payload (default: a few; a caller that saves several
    files side by side passes 1)."""
    ds = xp.unstack() if isinstance(xp, Dataset) else Dataset({xp.name or "data": xp})
    sizes = {k: _nbytes(v) for k, v in list(ds.data_vars.items()) + list(ds.coords.items())}
    biggest = max(sizes.values(), default=0)
    if shard_bytes is None and biggest <= _LIMIT:
        return _commit(file, [(str(file), ds)], threads)
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

    _commit(file, parts(), threads)


def _stale(file):
    import glob
    import os

    return [old for old in [str(file)] + glob.glob(glob.escape(str(file)) + ".part[0-9][0-9][0-9]") if os.path.isfile(old)]


def _commit(file, named, threads=None):
    """Write every (final name, dataset) of ``named`` under a temporary name (one at a time: a part comes to the host
    when it is written), then rename them all, then remove what an earlier save left that was not replaced."""
    import os
    import uuid

    tag = f".tmp{os.getpid()}-{uuid.uuid4().hex[:8]}"
    written = []
    try:
        for final, ds in named:
            _write_nc(final + tag, ds, threads)
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


_NC_TYPE = {"i1": 1, "S1": 2, "i2": 3, "i4": 4, "f4": 5, "f8": 6}  # NC_BYTE, NC_CHAR, NC_SHORT, NC_INT, NC_FLOAT, NC_DOUBLE


def _pad4(raw: bytes) -> bytes:
    return raw + b"\x00" * (-len(raw) % 4)


def _nc_name(name: str) -> bytes:
    raw = name.encode("utf-8")
    return _struct.pack(">i", len(raw)) + _pad4(raw)


def _nc_attrs(attrs: dict) -> bytes:
    """att_list: strings as NC_CHAR, Python / NumPy integers as NC_INT, floats as NC_DOUBLE (what scipy and xarray's
    scipy engine write for them)."""
    items = [(k, v) for k, v in attrs.items() if isinstance(v, (str, bytes, int, float, np.integer, np.floating))]
    if not items:
        return _struct.pack(">ii", 0, 0)  # ABSENT
    out = [_struct.pack(">ii", 0x0C, len(items))]
    for k, v in items:
        out.append(_nc_name(k))
        if isinstance(v, (str, bytes)):
            raw = v.encode("utf-8") if isinstance(v, str) else v
            out.append(_struct.pack(">ii", 2, len(raw)) + _pad4(raw))
        elif isinstance(v, (bool, int, np.integer)):
            out.append(_struct.pack(">iii", 4, 1, int(v)))
        else:
            out.append(_struct.pack(">iid", 6, 1, float(v)))
    return b"".join(out)


def _write_nc(file, ds, threads=None):
    """One NetCDF-3 file (64-bit offsets, no record dimension), written directly: the header from the classic format's
    grammar, then every variable's bytes at the offset the header names -- big-endian, as the format wants them.  An
    array that already IS big-endian (``dtype.byteorder == '>'``: a sink's staging block whose bytes were swapped on the
    device) or has one-byte items is written as it lies, in pieces, by the library's writer threads
    (``mg_host_write_runs``); anything else is swapped into a scratch buffer chunk by chunk.  ``mg.load`` (scipy's
    reader) and ``xarray.open_dataset`` read the result (same encoding conventions as before: ``_encode``)."""
    import os

    coord_names = [k for k, c in ds.coords.items() if not (c.dims == (k,))]
    dims, variables = {}, []  # name -> size; (name, dims, array, attrs)

    def put(name, v, is_data_var):
        arr, extra, attrs = _encode(name, np.asarray(v.values))
        vdims = tuple(v.dims) + extra
        for d, n in zip(vdims, arr.shape):
            if dims.setdefault(d, int(n)) != int(n):
                raise ValueError(f"{name}: dimension {d} has {n} entries here and {dims[d]} elsewhere")
        attrs = dict({k: a for k, a in dict(v.attrs or {}).items() if not k.startswith("__")}, **attrs)
        if is_data_var:
            mine = [c for c in coord_names if set(ds.coords[c].dims) <= set(v.dims)]
            if mine:
                attrs["coordinates"] = " ".join(mine)
        variables.append((name, vdims, arr, attrs))

    for k, v in ds.data_vars.items():
        put(k, v, True)
    for k, c in ds.coords.items():
        put(k, c, False)
    gattrs = {}
    if coord_names:
        gattrs["coordinates"] = " ".join(coord_names)
    for k, a in ds.attrs.items():
        if isinstance(a, (str, bytes, int, float, np.integer, np.floating)):
            gattrs[k] = a
        elif isinstance(a, (list, tuple)) and all(isinstance(x, str) for x in a):
            gattrs[k] = " ".join(a)
    dim_ids = {d: i for i, d in enumerate(dims)}

    def header(begins):
        out = [b"CDF\x02", _struct.pack(">i", 0)]
        if dims:
            out.append(_struct.pack(">ii", 0x0A, len(dims)))
            out += [_nc_name(d) + _struct.pack(">i", n) for d, n in dims.items()]
        else:
            out.append(_struct.pack(">ii", 0, 0))
        out.append(_nc_attrs(gattrs))
        if variables:
            out.append(_struct.pack(">ii", 0x0B, len(variables)))
            for (name, vdims, arr, attrs), begin in zip(variables, begins):
                kind = "S1" if arr.dtype.kind == "S" else f"{arr.dtype.kind}{arr.dtype.itemsize}"
                vsize = arr.nbytes + (-arr.nbytes % 4)
                out.append(_nc_name(name) + _struct.pack(">i", len(vdims)) + b"".join(_struct.pack(">i", dim_ids[d]) for d in vdims)
                           + _nc_attrs(attrs) + _struct.pack(">iIq", _NC_TYPE[kind], min(vsize, 0xFFFFFFFF), begin))
        else:
            out.append(_struct.pack(">ii", 0, 0))
        return b"".join(out)

    head_len = len(header([0] * len(variables)))
    begins, pos = [], head_len
    for _, _, arr, _ in variables:
        begins.append(pos)
        pos += arr.nbytes + (-arr.nbytes % 4)
    head = header(begins)
    assert len(head) == head_len
    fd = os.open(str(file), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.ftruncate(fd, pos)  # (the padding behind every variable reads as zeros)
        os.pwrite(fd, head, 0)
        runs, keep = [], []
        for (_, _, arr, _), begin in zip(variables, begins):
            if arr.nbytes == 0:
                continue
            if arr.dtype.itemsize > 1 and arr.dtype.byteorder != ">" and not (arr.dtype.byteorder == "=" and _sys.byteorder == "big"):
                arr = arr.astype(arr.dtype.newbyteorder(">"))  # (not a sink's block: swapped here)
            arr = np.ascontiguousarray(arr)
            keep.append(arr)
            flat = arr.reshape(-1).view(np.uint8)
            for lo in range(0, flat.size, _PIECE):
                runs.append((begin + lo, flat[lo: lo + _PIECE]))
        _write_runs(fd, runs, threads)
    finally:
        os.close(fd)


_PIECE = 8 << 20


def _write_runs(fd, runs, threads=None):
    """(file offset, uint8 array) pieces -> the file: by the library's writer threads when it is there, else one by one."""
    import os

    if not runs:
        return
    try:
        from . import _native

        lib = _native.lib()
    except Exception:  # noqa: BLE001  (mg.save on a machine without the built library: plain positional writes)
        lib = None
    if lib is None or len(runs) == 1:
        for off, piece in runs:
            view, done = memoryview(piece), 0
            while done < len(view):
                done += os.pwrite(fd, view[done:], off + done)
        return
    import ctypes

    n = len(runs)
    fds = np.full(n, fd, dtype=np.int32)
    offs = np.fromiter((r[0] for r in runs), dtype=np.int64, count=n)
    lens = np.fromiter((r[1].size for r in runs), dtype=np.int64, count=n)
    srcs = np.fromiter((r[1].ctypes.data for r in runs), dtype=np.uint64, count=n)
    failed = (ctypes.c_int64 * 2)(-1, 0)
    if threads is None:
        threads = max(1, min(8, (os.cpu_count() or 1) // 2))
    rc = lib.mg_host_write_runs(fds.ctypes.data, offs.ctypes.data, lens.ctypes.data, srcs.ctypes.data, n, int(threads),
                                ctypes.addressof(failed))
    if rc == -3:
        raise OSError(int(failed[1]), f"{os.strerror(int(failed[1])) if failed[1] else 'short write'} (NetCDF payload, run {int(failed[0])})")
    if rc != 0:
        raise RuntimeError(f"mg_host_write_runs failed ({rc})")


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
