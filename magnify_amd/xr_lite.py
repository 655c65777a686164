"""A small named-dimension container that mirrors the slice of the xarray API the hot path's
component interface uses (reference: every component is ``callable(xr.Dataset) -> xr.Dataset``,
registry.py:12-29).

xarray and dask are not installed in the build/GPU images, so the components exchange these
objects instead; ``to_xarray()`` converts a result into a real ``xarray.Dataset`` with the
reference's exact schema whenever xarray is importable.  Arrays may be NumPy
arrays or device-resident ``torch.Tensor``s (the "lazy" side of the reference's dask arrays:
they move to the host only when ``.values`` is read).
"""
from __future__ import annotations

import numpy as np

try:  # torch is optional for the container itself
    import torch
except Exception:  # pragma: no cover
    torch = None


def _is_tensor(x):
    return torch is not None and isinstance(x, torch.Tensor)


def _to_numpy(x):
    if _is_tensor(x):
        return x.detach().cpu().numpy()
    if hasattr(x, "materialize"):
        return _to_numpy(x.materialize())
    return np.asarray(x)


def _take(data, idx, axis):
    if not _is_tensor(data):
        return np.take(data, idx, axis=axis)
    index = torch.as_tensor(np.asarray(idx), device=data.device, dtype=torch.long)
    if data.dtype in (torch.uint16, torch.uint32, torch.uint64):  # index_select lacks unsigned kernels
        signed = {torch.uint16: torch.int16, torch.uint32: torch.int32, torch.uint64: torch.int64}[data.dtype]
        return data.view(signed).index_select(axis, index).view(data.dtype)
    return data.index_select(axis, index)


class DataArray:
    def __init__(self, data=None, dims=(), coords=None, name=None, attrs=None):
        if isinstance(dims, str):
            dims = (dims,)
        self._data = data if (_is_tensor(data) or hasattr(data, "materialize")) else np.asarray(data)
        self.dims = tuple(dims)
        if len(self.dims) != len(self.shape):
            raise ValueError(f"dims {self.dims} do not match data of shape {self.shape}")
        self.coords = {}
        for k, v in (coords or {}).items():
            self.coords[k] = v if isinstance(v, DataArray) else DataArray(np.asarray(v), (k,))
        self.name = name
        self.attrs = dict(attrs or {})

    # -- basic properties ------------------------------------------------------------------
    @property
    def data(self):
        if hasattr(self._data, "materialize"):
            self._data = self._data.materialize()
        return self._data

    @data.setter
    def data(self, value):
        self._data = value

    @property
    def raw(self):
        """The backing object without forcing a lazy operand."""
        return self._data

    @property
    def shape(self):
        return tuple(self._data.shape)

    @property
    def dtype(self):
        d = self._data.dtype
        return np.dtype(str(d).replace("torch.", "")) if _is_tensor(self._data) or hasattr(self._data, "materialize") else d

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def sizes(self):
        return dict(zip(self.dims, self.shape))

    @property
    def values(self):
        return _to_numpy(self.data)

    def to_numpy(self):
        return self.values

    def item(self):
        return self.values.item()

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        v = self.values
        return v.astype(dtype) if dtype is not None else v

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def __repr__(self):
        return f"<magnify_amd.DataArray {self.name or ''} {self.sizes} {self.dtype}>"

    def __getattr__(self, name):
        coords = self.__dict__.get("coords", {})
        if name in coords:
            return coords[name]
        raise AttributeError(name)

    # -- structure ---------------------------------------------------------------------------
    def _replace(self, data, dims, coords=None):
        if coords is None:
            coords = {k: v for k, v in self.coords.items() if set(v.dims) <= set(dims)}
        return DataArray(data, dims, coords, self.name, self.attrs)

    def copy(self):
        return self._replace(self.data, self.dims, dict(self.coords))

    def assign_attrs(self, attrs=None, **kw):
        out = self.copy()
        out.attrs.update(attrs or {})
        out.attrs.update(kw)
        return out

    # -- chunk policy (the reference's dask chunking of roi / fg / bg, find.py:63-88, 185-201, 506-531) ----------------
    def chunk(self, sizes=None, **kw):
        """Record a chunk size per dimension (``xarray.DataArray.chunk``).  Nothing is cut here -- the data stays one
        resident array -- but the policy travels with the variable: ``to_xarray()`` hands it to dask when that is
        installed, and ``mg.save`` cuts its parts at chunk boundaries."""
        out = self.copy()
        have = dict(out.attrs.get("__mg_chunks__", {}))
        have.update({d: int(n) for d, n in dict(sizes or {}, **kw).items() if d in out.dims})
        out.attrs["__mg_chunks__"] = have
        return out

    @property
    def chunksizes(self):
        """{dim: chunk lengths} as xarray reports them, or {} when no policy is attached."""
        pol = self.attrs.get("__mg_chunks__")
        if not pol:
            return {}
        out = {}
        for d, n in self.sizes.items():
            c = max(1, min(int(pol.get(d, n)), n)) if n else 1
            out[d] = tuple([c] * (n // c) + ([n % c] if n % c else [])) if n else (0,)
        return out

    @property
    def chunks(self):
        cs = self.chunksizes
        return tuple(cs[d] for d in self.dims) if cs else None

    def rename(self, mapping=None, **kw):
        mapping = dict(mapping or {}, **kw)
        dims = tuple(mapping.get(d, d) for d in self.dims)
        coords = {mapping.get(k, k): v.rename(mapping) if any(d in mapping for d in v.dims) else v
                  for k, v in self.coords.items()}
        return DataArray(self._data, dims, coords, self.name, self.attrs)

    def transpose(self, *dims):
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            rest = [d for d in self.dims if d not in dims]
            dims = tuple(dims[:i]) + tuple(rest) + tuple(dims[i + 1:])
        dims = tuple(d for d in dims if d in self.dims)
        perm = [self.dims.index(d) for d in dims]
        if perm == list(range(self.ndim)):  # already in order: a pending (lazy) operand stays pending
            return self._replace(self._data, dims, dict(self.coords))
        data = self.data.permute(*perm) if _is_tensor(self.data) else np.transpose(self.data, perm)
        return self._replace(data, dims, dict(self.coords))

    def expand_dims(self, dim, axis=0):
        data = self.data.unsqueeze(axis) if _is_tensor(self.data) else np.expand_dims(self.data, axis)
        dims = self.dims[:axis] + (dim,) + self.dims[axis:]
        return self._replace(data, dims, dict(self.coords))

    def squeeze(self, dim=None):
        """Drop dimensions of size 1 (all of them, one, or the listed ones): one reshape of the data and of the coords
        that carry such a dimension (they stay as lower-dimensional / scalar coords, as after ``isel({d: 0})``)."""
        sizes = self.sizes
        dims = [d for d in self.dims if sizes[d] == 1] if dim is None else ([dim] if isinstance(dim, str) else list(dim))
        for d in dims:
            if sizes[d] != 1:
                raise ValueError(f"cannot squeeze dimension {d} of size {sizes[d]}")
        if not dims:
            return self
        gone = set(dims)
        keep = tuple(d for d in self.dims if d not in gone)
        data = self.data
        data = data.reshape(tuple(sizes[d] for d in keep))
        coords = {}
        for name, c in self.coords.items():
            if all(cd in sizes for cd in c.dims):
                if gone.isdisjoint(c.dims):
                    coords[name] = DataArray(c.raw, c.dims, None, c.name, c.attrs)
                else:
                    cdims = tuple(cd for cd in c.dims if cd not in gone)
                    cdata = c.data
                    coords[name] = DataArray(cdata.reshape(tuple(s for cd, s in zip(c.dims, c.shape) if cd not in gone)), cdims,
                                             None, c.name, c.attrs)
        return DataArray(data, keep, coords, self.name, self.attrs)

    def isel(self, indexers=None, **kw):
        indexers = dict(indexers or {}, **kw)
        key = tuple(indexers.get(d, slice(None)) for d in self.dims)
        return self[key]

    def sel(self, indexers=None, **kw):
        indexers = dict(indexers or {}, **kw)
        pos = {}
        for d, labels in indexers.items():
            ticks = list(self.coords[d].values.tolist()) if d in self.coords else list(range(self.sizes[d]))
            if isinstance(labels, DataArray):
                labels = labels.values.tolist()
            if isinstance(labels, (list, tuple, np.ndarray)):
                pos[d] = [ticks.index(v) for v in (labels.tolist() if isinstance(labels, np.ndarray) else labels)]
            else:
                pos[d] = ticks.index(labels)
        return self.isel(pos)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.coords[key]
        if not isinstance(key, tuple):
            key = (key,)
        if Ellipsis in key:
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (self.ndim - len(key) + 1) + key[i + 1:]
        key = key + (slice(None),) * (self.ndim - len(key))
        norm = key
        # apply the indexers one axis at a time (orthogonal indexing, as xarray)
        out = self.data
        dims = []
        axis = 0
        for d, k in zip(self.dims, norm):
            if isinstance(k, (int, np.integer)):
                out = out.select(axis, int(k)) if _is_tensor(out) else np.take(out, int(k), axis=axis)
            elif isinstance(k, slice):
                if k != slice(None):  # (a full slice is the array itself: no indexing call)
                    out = out[(slice(None),) * axis + (k,)]
                dims.append(d)
                axis += 1
            else:
                idx = np.asarray(k)
                if idx.dtype == bool:
                    idx = np.nonzero(idx)[0]
                out = _take(out, idx, axis)
                dims.append(d)
                axis += 1
        coords = {}
        for name, c in self.coords.items():
            if all(cd in self.dims for cd in c.dims):
                ckey = tuple(norm[self.dims.index(cd)] for cd in c.dims)
                bare = DataArray(c.raw, c.dims, None, c.name, c.attrs)
                sub = bare[ckey] if c.ndim else bare
                if set(sub.dims) <= set(dims):
                    coords[name] = sub
        return DataArray(out, tuple(dims), coords, self.name, self.attrs)

    def __setitem__(self, key, value):
        if isinstance(value, DataArray):
            value = value.data
        data = self.data
        if _is_tensor(data):
            if not _is_tensor(value):
                value = torch.as_tensor(np.asarray(value), device=data.device)
            data[key] = value.to(data.dtype)
        else:
            data[key] = _to_numpy(value)

    # -- arithmetic on the host (user-side algebra such as README.md:21-22) --------------------
    def _binary(self, other, op):
        if isinstance(other, DataArray):
            dims = list(self.dims) + [d for d in other.dims if d not in self.dims]
            a = _align(self, dims)
            b = _align(other, dims)
            coords = dict(other.coords)
            coords.update(self.coords)
            return DataArray(op(a, b), dims, {k: v for k, v in coords.items() if set(v.dims) <= set(dims)}, self.name)
        return self._replace(op(self.values, other), self.dims, dict(self.coords))

    def __add__(self, o): return self._binary(o, np.add)
    def __sub__(self, o): return self._binary(o, np.subtract)
    def __mul__(self, o): return self._binary(o, np.multiply)
    def __truediv__(self, o): return self._binary(o, np.divide)
    def __and__(self, o): return self._binary(o, np.logical_and)
    def __or__(self, o): return self._binary(o, np.logical_or)
    def __invert__(self): return self._replace(~self.values, self.dims, dict(self.coords))
    def __eq__(self, o): return self._binary(o, np.equal)  # noqa: E704
    def __ne__(self, o): return self._binary(o, np.not_equal)
    def __lt__(self, o): return self._binary(o, np.less)
    def __gt__(self, o): return self._binary(o, np.greater)
    def __le__(self, o): return self._binary(o, np.less_equal)
    def __ge__(self, o): return self._binary(o, np.greater_equal)
    __hash__ = None

    def astype(self, dtype):
        return self._replace(self.values.astype(dtype), self.dims, dict(self.coords))

    def where(self, cond):
        """NaN outside ``cond`` (float64), broadcasting by dimension name."""
        if isinstance(cond, str):
            cond = self.coords[cond]
        dims = list(self.dims) + [d for d in cond.dims if d not in self.dims]
        a = _align(self, dims).astype(np.float64)
        c = _align(cond, dims).astype(bool)
        coords = dict(cond.coords)
        coords.update(self.coords)
        return DataArray(np.where(c, a, np.nan), dims, {k: v for k, v in coords.items() if set(v.dims) <= set(dims)},
                         self.name)

    def _reduce(self, fn, dim=None, **kw):
        if dim is None:
            axes, dims = None, ()
        else:
            names = [dim] if isinstance(dim, str) else list(dim)
            axes = tuple(self.dims.index(d) for d in names)
            dims = tuple(d for d in self.dims if d not in names)
        return self._replace(fn(self.values, axis=axes, **kw), dims)

    def sum(self, dim=None): return self._reduce(np.nansum if self.dtype.kind == "f" else np.sum, dim)
    def mean(self, dim=None): return self._reduce(np.nanmean, dim)
    def median(self, dim=None): return self._reduce(np.nanmedian, dim)
    def max(self, dim=None): return self._reduce(np.nanmax if self.dtype.kind == "f" else np.max, dim)
    def min(self, dim=None): return self._reduce(np.nanmin if self.dtype.kind == "f" else np.min, dim)
    def any(self, dim=None): return self._reduce(np.any, dim)
    def all(self, dim=None): return self._reduce(np.all, dim)

    def assign_coords(self, coords=None, **kw):
        out = self.copy()
        for k, v in dict(coords or {}, **kw).items():
            out.coords[k] = _as_coord(k, v)
        return out

    def to_xarray(self):
        import xarray as xr

        coords = {k: (v.dims, v.values) for k, v in self.coords.items()}
        attrs = {k: v for k, v in self.attrs.items() if not k.startswith("__mg")}
        out = xr.DataArray(self.values, dims=self.dims, coords=coords, name=self.name, attrs=attrs)
        return _xr_chunk(out, self.attrs.get("__mg_chunks__"))


def _align(arr: DataArray, dims):
    """NumPy view of ``arr`` broadcastable against ``dims``."""
    v = arr.values
    have = [d for d in dims if d in arr.dims]
    v = np.transpose(v, [arr.dims.index(d) for d in have])
    shape = [arr.sizes[d] if d in arr.dims else 1 for d in dims]
    return v.reshape(shape)


def _as_coord(name, v):
    if isinstance(v, DataArray):
        return v
    if isinstance(v, tuple) and len(v) == 2:
        dims, data = v
        return DataArray(data, dims, name=name)
    return DataArray(np.asarray(v), (name,), name=name)


class Dataset:
    """Dict of named variables + coordinates sharing dimension names."""

    def __init__(self, data_vars=None, coords=None, attrs=None):
        object.__setattr__(self, "data_vars", {})
        object.__setattr__(self, "coords", {})
        object.__setattr__(self, "attrs", dict(attrs or {}))
        object.__setattr__(self, "_cache", {})
        for k, v in (coords or {}).items():
            self.coords[k] = _as_coord(k, v)
        for k, v in (data_vars or {}).items():
            self[k] = v

    @property
    def mg(self):
        """``Dataset.mg.cache(vars)`` of the reference (accessor.py:18-35) spills lazy dask arrays to
        zarr; here arrays are already resident in HBM, so caching only materialises pending
        (lazy flat-field) operands."""
        return _Accessor(self)

    # -- mapping protocol ---------------------------------------------------------------------
    def __contains__(self, name):
        return name in self.data_vars or name in self.coords

    def __getitem__(self, name):
        if name in self.data_vars:
            return self._with_coords(self.data_vars[name])
        if name in self.coords:
            return self._with_coords(self.coords[name])
        raise KeyError(name)

    def _with_coords(self, arr):
        """A view of ``arr`` carrying the dataset's matching coordinates (bare: no nested coords)."""
        coords = {k: DataArray(c.raw, c.dims, None, k, c.attrs) for k, c in self.coords.items()
                  if c is not arr and set(c.dims) <= set(arr.dims)}
        return DataArray(arr.raw, arr.dims, coords, arr.name, arr.attrs)

    def __setitem__(self, name, value):
        if isinstance(value, tuple) and len(value) == 2:
            value = DataArray(value[1], value[0])
        if not isinstance(value, DataArray):
            raise TypeError("Dataset values must be DataArray or (dims, data)")
        value = DataArray(value.raw, value.dims, None, name, value.attrs)
        for d, n in value.sizes.items():
            if d in self.sizes and self.sizes[d] != n:
                raise ValueError(f"conflicting sizes for dimension {d!r}: {self.sizes[d]} vs {n}")
        self.data_vars[name] = value

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__") and name not in ("__original_tile_dims__",):
            raise AttributeError(name)
        if name in self.data_vars or name in self.coords:
            return self[name]
        if name in self.attrs:
            return self.attrs[name]
        raise AttributeError(f"Dataset has no variable, coordinate or attribute {name!r}")

    def __setattr__(self, name, value):
        raise AttributeError("use ds[name] = ... or assign_coords")

    @property
    def variables(self):
        out = dict(self.coords)
        out.update(self.data_vars)
        return out

    @property
    def sizes(self):
        out = {}
        for v in list(self.coords.values()) + list(self.data_vars.values()):
            out.update(v.sizes)
        return out

    @property
    def dims(self):
        return self.sizes

    def __repr__(self):
        dv = ", ".join(f"{k}{list(v.dims)}" for k, v in self.data_vars.items())
        co = ", ".join(f"{k}{list(v.dims)}" for k, v in self.coords.items())
        return f"<magnify_amd.Dataset sizes={self.sizes} data_vars=({dv}) coords=({co})>"

    # -- structure ------------------------------------------------------------------------------
    def copy(self):
        out = Dataset(attrs=self.attrs)
        out.coords.update(self.coords)
        out.data_vars.update(self.data_vars)
        out._cache.update(self._cache)
        return out

    def assign_attrs(self, attrs=None, **kw):
        out = self.copy()
        out.attrs.update(attrs or {})
        out.attrs.update(kw)
        return out

    # -- chunk policy (the reference's dask chunking of roi / fg / bg, find.py:63-88, 185-201, 506-531) ----------------
    def chunk(self, sizes=None, **kw):
        """Record a chunk size per dimension (``xarray.DataArray.chunk``).  Nothing is cut here -- the data stays one
        resident array -- but the policy travels with the variable: ``to_xarray()`` hands it to dask when that is
        installed, and ``mg.save`` cuts its parts at chunk boundaries."""
        out = self.copy()
        have = dict(out.attrs.get("__mg_chunks__", {}))
        have.update({d: int(n) for d, n in dict(sizes or {}, **kw).items() if d in out.dims})
        out.attrs["__mg_chunks__"] = have
        return out

    @property
    def chunksizes(self):
        """{dim: chunk lengths} as xarray reports them, or {} when no policy is attached."""
        pol = self.attrs.get("__mg_chunks__")
        if not pol:
            return {}
        out = {}
        for d, n in self.sizes.items():
            c = max(1, min(int(pol.get(d, n)), n)) if n else 1
            out[d] = tuple([c] * (n // c) + ([n % c] if n % c else [])) if n else (0,)
        return out

    @property
    def chunks(self):
        cs = self.chunksizes
        return tuple(cs[d] for d in self.dims) if cs else None

    def assign_coords(self, coords=None, **kw):
        out = self.copy()
        for k, v in dict(coords or {}, **kw).items():
            c = _as_coord(k, v)
            c.name = k
            out.coords[k] = c
        return out

    def drop_vars(self, names, errors="raise"):
        names = [names] if isinstance(names, str) else list(names)
        out = self.copy()
        for n in names:
            if n in out.data_vars:
                del out.data_vars[n]
            elif n in out.coords:
                del out.coords[n]
            elif errors == "raise":
                raise ValueError(f"variable {n!r} not found")
        return out

    def _map(self, fn, only=None):
        out = Dataset(attrs=self.attrs)
        out._cache.update(self._cache)
        for k, v in self.coords.items():
            out.coords[k] = fn(v) if (only is None or only(v)) else v
        for k, v in self.data_vars.items():
            nv = fn(v) if (only is None or only(v)) else v
            nv.name = k
            out.data_vars[k] = nv
        return out

    def rename(self, mapping=None, **kw):
        mapping = dict(mapping or {}, **kw)
        out = self._map(lambda v: v.rename(mapping))
        for old, new in mapping.items():
            if old in out.coords:
                out.coords[new] = out.coords.pop(old)
            if old in out.data_vars:
                out.data_vars[new] = out.data_vars.pop(old)
        return out

    def transpose(self, *dims):
        return self._map(lambda v: v.transpose(*[d for d in dims if d is Ellipsis or d in v.dims]) if v.ndim > 1 else v)

    def squeeze(self, dim):
        return self._map(lambda v: v.squeeze(dim), only=lambda v: dim in v.dims)

    def isel(self, indexers=None, **kw):
        indexers = dict(indexers or {}, **kw)
        return self._map(lambda v: v.isel({d: i for d, i in indexers.items() if d in v.dims}),
                         only=lambda v: any(d in v.dims for d in indexers))

    def sel(self, indexers=None, **kw):
        indexers = dict(indexers or {}, **kw)
        pos = {}
        for d, labels in indexers.items():
            ticks = self.coords[d].values.tolist() if d in self.coords else list(range(self.sizes[d]))
            pos[d] = [ticks.index(v) for v in labels] if isinstance(labels, (list, tuple, np.ndarray)) else ticks.index(labels)
        return self.isel(pos)

    def where(self, cond):
        if isinstance(cond, str):
            cond = self[cond]
        return self._map(lambda v: v.where(cond), only=lambda v: v.name in self.data_vars)

    # -- mark <-> (mark_row, mark_col) ---------------------------------------------------------------
    def stack_mark(self):
        """stack(mark=("mark_row", "mark_col")).transpose("mark", ...) (find.py:182)."""
        nr, nc = self.sizes["mark_row"], self.sizes["mark_col"]

        def fn(v):
            v = v.transpose("mark_row", "mark_col", ...)
            data = v.data.reshape((nr * nc,) + tuple(v.shape[2:]))
            return DataArray(data, ("mark",) + v.dims[2:], None, v.name, v.attrs)

        out = self._map(fn, only=lambda v: "mark_row" in v.dims and "mark_col" in v.dims)
        rows = out.coords.pop("mark_row", DataArray(np.arange(nr), ("mark_row",))).values
        cols = out.coords.pop("mark_col", DataArray(np.arange(nc), ("mark_col",))).values
        out.coords["mark_row"] = DataArray(np.repeat(rows, nc), ("mark",), name="mark_row")
        out.coords["mark_col"] = DataArray(np.tile(cols, nr), ("mark",), name="mark_col")
        out._cache["mark_shape"] = (nr, nc)
        return out

    def unstack(self):
        """Undo stack_mark (and nothing else): mark -> (mark_row, mark_col) (postprocess.py:21)."""
        if "mark_shape" not in self._cache or "mark" not in self.sizes:
            return self.copy()
        nr, nc = self._cache["mark_shape"]
        rows = self.coords["mark_row"].values.reshape(nr, nc)[:, 0]
        cols = self.coords["mark_col"].values.reshape(nr, nc)[0]

        def fn(v):
            ax = v.dims.index("mark")
            v = v.transpose("mark", ...)
            data = v.data.reshape((nr, nc) + tuple(v.shape[1:]))
            return DataArray(data, ("mark_row", "mark_col") + v.dims[1:], None, v.name, v.attrs)

        out = self.drop_vars(["mark_row", "mark_col"])
        out = out._map(fn, only=lambda v: "mark" in v.dims)
        out.coords["mark_row"] = DataArray(rows, ("mark_row",), name="mark_row")
        out.coords["mark_col"] = DataArray(cols, ("mark_col",), name="mark_col")
        del out._cache["mark_shape"]
        return out

    def to_xarray(self):
        """A real xarray.Dataset with the reference's schema (needs xarray)."""
        import xarray as xr

        def var(v):
            return (v.dims, v.values, {k: a for k, a in v.attrs.items() if not k.startswith("__mg")})  # (dims, data, attrs)

        coords = {k: var(v) for k, v in self.coords.items()}
        data_vars = {k: var(v) for k, v in self.data_vars.items()}
        attrs = {k: v for k, v in self.attrs.items() if not k.startswith("__mg")}
        ds = xr.Dataset(data_vars, coords=coords, attrs=attrs)
        for k, v in list(self.data_vars.items()) + list(self.coords.items()):  # the reference's dask chunks, if dask is here
            if v.attrs.get("__mg_chunks__"):
                ds[k] = _xr_chunk(ds[k], v.attrs["__mg_chunks__"])
        if "mark_shape" in self._cache and "mark" in self.sizes:
            ds = ds.set_index(mark=("mark_row", "mark_col"))
        return ds


def _xr_chunk(obj, policy):
    if not policy:
        return obj
    try:
        import dask  # noqa: F401
    except Exception:  # no dask: a NumPy-backed xarray object (chunking is a property of the lazy backend)
        return obj
    return obj.chunk({d: n for d, n in policy.items() if d in obj.dims})


_SPILL_DIRS = []  # TemporaryDirectory objects of disk spills, alive as long as the process (accessor.py:8, 13-16)


class _Accessor:
    """``Dataset.mg`` (reference: src/magnify/accessor.py:11-35).  The reference's ``cache`` writes lazy dask arrays to
    a zarr store in a temporary directory and swaps the stored array in -- how it holds "terabytes on a laptop".  Here
    the arrays are device-resident tensors: ``cache()`` forces pending operands and keeps them in HBM (288 GB);
    ``cache(spill="host")`` moves the named variables to page-locked host memory, ``cache(spill="disk")`` (or a
    directory path) to memory-mapped ``.npy`` files in a temporary directory that lives as long as the process -- the
    swap-in-place the reference does, with the same call."""

    def __init__(self, ds):
        self._ds = ds

    def cache(self, variables=None, spill=None):
        names = [variables] if isinstance(variables, str) else list(variables or self._ds.variables)
        if spill is None:
            import os

            spill = os.environ.get("MG_CACHE_SPILL") or None
        for n in names:
            arr = self._ds.variables[n]
            data = arr.data  # forces a pending operand
            if spill is None or not _is_tensor(data) or not data.is_cuda:
                continue
            if spill == "host":
                host = torch.empty(data.shape, dtype=data.dtype).pin_memory() if data.numel() else torch.empty(data.shape, dtype=data.dtype)
                host.copy_(data)  # (also materialises expanded views: one copy per variable)
                arr.data = host.numpy()
            else:
                import os
                import tempfile

                if spill == "disk":
                    if not _SPILL_DIRS:
                        _SPILL_DIRS.append(tempfile.TemporaryDirectory(prefix="magnify_amd_cache_"))
                    root = _SPILL_DIRS[0].name
                else:
                    root = str(spill)
                    os.makedirs(root, exist_ok=True)
                import uuid

                # (a name of its own: `id(ds)` comes back after a dataset is collected -- "w+" on a recycled name would
                # truncate a file an older array still has memory-mapped)
                path = os.path.join(root, f"{uuid.uuid4().hex}_{n}.npy")
                host = _to_numpy(data)
                mm = np.lib.format.open_memmap(path, mode="w+", dtype=host.dtype, shape=host.shape)
                mm[...] = host
                mm.flush()
                arr.data = np.load(path, mmap_mode="r+")
        return self._ds


def from_any(obj):
    """Accept magnify_amd or real xarray DataArray/Dataset objects."""
    if isinstance(obj, (DataArray, Dataset)):
        return obj
    mod = type(obj).__module__
    if mod.startswith("xarray"):
        if hasattr(obj, "data_vars"):
            ds = Dataset(attrs=dict(obj.attrs))
            for k, v in obj.coords.items():
                ds.coords[k] = DataArray(np.asarray(v.values), v.dims, name=k)
            for k, v in obj.data_vars.items():
                ds[k] = DataArray(np.asarray(v.values), v.dims)
            return ds
        coords = {k: DataArray(np.asarray(v.values), v.dims, name=k) for k, v in obj.coords.items()}
        return DataArray(np.asarray(obj.values), obj.dims, coords, obj.name, dict(obj.attrs))
    raise TypeError(f"expected a DataArray/Dataset, got {type(obj)}")
