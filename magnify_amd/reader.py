"""Reader (reference: src/magnify/reader.py).

In-memory inputs (DataArray / Dataset objects, magnify_amd's or real xarray's, or a sequence of
them) pass straight through (reader.py:31-35).  Path patterns follow the reference's grammar
(reader.py:80-160): named groups ``(assay)``, ``(channel)``, ``(time)`` / ``(time|%Y%m%d)``,
``(row)``, ``(col)`` select the file's place in the tile array, ``(name_key|formatter|format)`` groups
attach an alternative labelling ``name`` to dimension ``key``; ``*`` / ``?`` / ``**`` glob as usual.
Files are TIFFs read with Pillow (tifffile / OME / MicroManager metadata are not available to this
build): one 2-D page per file, or an ImageJ hyperstack whose description names ``channels`` /
``frames``.  Reading is eager into one host array per assay -- SURVEY 8f N2's streaming half is the
per-stream upload of ``StackProcessor`` (host-resident stack -> HBM, overlapped with compute).
"""
from __future__ import annotations

import collections
import datetime
import fnmatch
import glob
import os
import re

import numpy as np

from . import registry, xr_lite
from .utils import natural_sort_key

_GROUP = re.compile(r"\(([^()|]*?)(?:\s*\|\s*([^()|]*?))?(?:\s*\|\s*([^()|]*?))?\)")
_FORMAT = {
    "": lambda text, fmt: text,
    "str": lambda text, fmt: text,
    "int": lambda text, fmt: int(text),
    "float": lambda text, fmt: float(text),
    "time": lambda text, fmt: datetime.datetime.strptime(text, fmt if fmt else "%Y%m%d-%H%M%S"),
}


def _glob_to_regex(literal: str) -> str:
    """fnmatch's translation of a literal pattern piece, without its anchors."""
    full = fnmatch.translate(literal)
    return full[len("(?s:") : full.rindex(")")]


def extract_paths(pattern, **keys):
    """reader.py:80-160.  Returns (path_dict, meta_dict): ``path_dict[idx tuple over all keys] = path``
    (None where the pattern has no group for a key), ``meta_dict[(name, key)][key value] = meta value``."""
    pattern = os.path.expanduser(str(pattern))
    kinds = {k: (f if callable(f) else _FORMAT[f]) for k, f in keys.items()}
    glob_parts, regex_parts, pos = [], [], 0
    dim_fmt, metas = {}, []  # key -> format string;  (name, key, formatter, format string)
    for m in _GROUP.finditer(pattern):
        head, a, b = m.group(1).strip(), m.group(2), m.group(3)
        owner = next((k for k in keys if head == k), None)
        meta_of = next((k for k in keys if head.endswith("_" + k) and len(head) > len(k) + 1), None)
        if owner is None and meta_of is None:
            continue  # an ordinary parenthesis in a file name
        glob_parts.append(pattern[pos : m.start()] + "*")
        regex_parts.append(_glob_to_regex(pattern[pos : m.start()]))
        if owner is not None:
            dim_fmt[owner] = a
            regex_parts.append(f"(?P<{owner}>[^/\\\\]*?)")
        else:
            name = head[: -len(meta_of) - 1]
            metas.append((name, meta_of, _FORMAT[(a or "").strip()], b))
            regex_parts.append(f"(?P<{name}>[^/\\\\]*?)")
        pos = m.end()
    glob_parts.append(pattern[pos:])
    regex_parts.append(_glob_to_regex(pattern[pos:]))
    regex = re.compile("(?s:" + "".join(regex_parts) + r")\Z", re.IGNORECASE)
    path_dict, meta_dict = {}, collections.defaultdict(dict)
    for path in glob.glob("".join(glob_parts), recursive=True):
        match = regex.fullmatch(path)
        if match is None:
            continue
        idx = tuple(kinds[k](match.group(k), dim_fmt[k]) if k in dim_fmt else None for k in keys)
        for name, key, formatter, fmt in metas:
            if key in dim_fmt:
                meta_dict[name, key][idx[list(keys).index(key)]] = formatter(match.group(name), fmt)
        if idx in path_dict:
            raise ValueError(f"{path} and {path_dict[idx]} map to the same index.")
        path_dict[idx] = os.path.abspath(path)
    return path_dict, meta_dict


def _open_tiff(path):
    """-> (pages as a list of 2-D arrays, dims of the page axis: [] / ['time'] / ['channel'] / both)."""
    from PIL import Image

    with Image.open(path) as im:
        n = getattr(im, "n_frames", 1)
        desc = im.tag_v2.get(270, "") if hasattr(im, "tag_v2") else ""
        pages = []
        for i in range(n):
            im.seek(i)
            pages.append(np.array(im))
    if n == 1:
        return pages, [], ()
    found = {k: int(v) for k, v in re.findall(r"(channels|frames|slices)=(\d+)", desc if isinstance(desc, str) else "")}
    if found.get("slices", 1) > 1:
        raise ValueError("tiff files with a Z dimension are not yet supported.")
    n_c, n_t = found.get("channels", 1), found.get("frames", 1)
    if n_c * n_t != n:
        raise ValueError(f"{path}: {n} pages but no ImageJ hyperstack description that explains them "
                         "(OME / MicroManager metadata are not readable in this build)")
    dims, shape = [], ()
    if n_t > 1:
        dims, shape = dims + ["time"], shape + (n_t,)
    if n_c > 1:
        dims, shape = dims + ["channel"], shape + (n_c,)
    return pages, dims, shape


def read_tiffs(xp_dict, name, meta_dict):
    """reader.py:163-324: one Dataset with ``tile`` over the dimensions found in the paths and inside
    the files, in the standard order (channel, time, tile_row, tile_col, tile_y, tile_x)."""
    channel_idx, time_idx, row_idx, col_idx = (sorted(set(i)) for i in zip(*xp_dict.keys()))
    in_path, outer = [], ()
    for dim, values in (("channel", channel_idx), ("time", time_idx), ("tile_row", row_idx), ("tile_col", col_idx)):
        if values[0] != -1:
            in_path.append(dim)
            outer += (len(values),)
    files = [p for _, p in sorted(xp_dict.items())]
    pages0, in_file, inner = _open_tiff(files[0])
    if set(in_file) & set(in_path):
        raise ValueError("Dimensions specified in the path names and inside the tiff file overlap.")
    ty, tx = pages0[0].shape[:2]
    if len(files) != int(np.prod(outer, dtype=np.int64)):
        raise ValueError(f"{name or 'assay'}: {len(files)} files do not fill the {outer} array the pattern describes")
    tiles = np.empty(outer + inner + (ty, tx), dtype=pages0[0].dtype)
    flat = tiles.reshape((-1,) + (ty, tx))
    per_file = int(np.prod(inner, dtype=np.int64)) if inner else 1
    for f, path in enumerate(files):
        pages = pages0 if f == 0 else _open_tiff(path)[0]
        if len(pages) != per_file or pages[0].shape[:2] != (ty, tx):
            raise ValueError(f"{path}: page count / size differs from {files[0]}")
        for k, page in enumerate(pages):
            flat[f * per_file + k] = page
    coords = {}
    if "channel" in in_path:
        coords["channel"] = list(channel_idx)
    if "time" in in_path:
        coords["time"] = [int(t.timestamp()) if isinstance(t, datetime.datetime) else t for t in time_idx]
    xp = xr_lite.Dataset({"tile": xr_lite.DataArray(tiles, tuple(in_path + in_file + ["tile_y", "tile_x"]))},
                         coords=coords, attrs={"name": name})
    order = [d for d in ("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x") if d in xp.tile.dims]
    xp = xp.transpose(*order)
    for (meta_name, dim), table in meta_dict.items():
        keys = time_idx if dim == "time" else {"channel": channel_idx, "row": row_idx, "col": col_idx}.get(dim, [])
        axis = {"row": "tile_row", "col": "tile_col"}.get(dim, dim)
        if axis in xp.sizes:
            xp = xp.assign_coords({meta_name: ((axis,), [table[k] for k in keys])})
    return xp


class Reader:
    def __call__(self, data):
        single = isinstance(data, (str, bytes, os.PathLike, xr_lite.DataArray, xr_lite.Dataset)) or \
            type(data).__module__.startswith("xarray")
        for d in ([data] if single else data):
            if not isinstance(d, (str, bytes, os.PathLike)):
                yield xr_lite.from_any(d)
                continue
            path_dict, meta_dict = extract_paths(os.fspath(d), assay="str", channel="str", time="time", row="int",
                                                 col="int")
            if len(path_dict) == 0:
                raise FileNotFoundError(f"The pattern {d} did not lead to any files.")
            path_dict = {(("",) + k[1:]) if k[0] is None else k: v for k, v in path_dict.items()}
            for xp_name in sorted({k[0] for k in path_dict}, key=natural_sort_key):
                xp_dict = {tuple(-1 if x is None else x for x in k[1:]): v for k, v in path_dict.items()
                           if k[0] == xp_name}
                yield read_tiffs(xp_dict, name=xp_name, meta_dict=meta_dict)

    @registry.readers.register("read")
    def make():
        return Reader()


def _tiff_layout(path):
    """(page dims, page shape, page dtype, n_frames) of a TIFF without decoding more than its first page."""
    from PIL import Image

    with Image.open(path) as im:
        n = getattr(im, "n_frames", 1)
        desc = im.tag_v2.get(270, "") if hasattr(im, "tag_v2") else ""
        first = np.array(im)
    if n == 1:
        return [], (), first.shape[:2], first.dtype, 1
    found = {k: int(v) for k, v in re.findall(r"(channels|frames|slices)=(\d+)", desc if isinstance(desc, str) else "")}
    if found.get("slices", 1) > 1:
        raise ValueError("tiff files with a Z dimension are not yet supported.")
    n_c, n_t = found.get("channels", 1), found.get("frames", 1)
    if n_c * n_t != n:
        raise ValueError(f"{path}: {n} pages but no ImageJ hyperstack description that explains them")
    dims, shape = [], ()
    if n_t > 1:
        dims, shape = dims + ["time"], shape + (n_t,)
    if n_c > 1:
        dims, shape = dims + ["channel"], shape + (n_c,)
    return dims, shape, first.shape[:2], first.dtype, n


def _read_page(path, index, out):
    """Decode ONE page of a TIFF into ``out`` (the reference maps every page to its own dask block and reads it
    on demand, reader.py:265-292)."""
    from PIL import Image

    with Image.open(path) as im:
        if index:
            im.seek(index)
        out[...] = np.asarray(im)


def iter_time_chunks(pattern, chunk: int, pinned: bool = False):
    """Streamed ingest of a time series too large to hold: the files behind
    ``pattern`` are read ``chunk`` timepoints at a time, in time order, page by page, never all at once.
    Groups as in ``extract_paths``: ``(channel)``, ``(time|format)``, and for tiled acquisitions ``(row)`` /
    ``(col)``; a file holds one 2-D page or an ImageJ hyperstack whose pages run over time and / or channel
    (a dimension is either in the path or in the file, reader.py:225-231).
    Yields ``(time_values, channels, block)`` with ``block`` (T_chunk, C, H, W) -- tiled series:
    (T_chunk, C, rows, cols, tile_y, tile_x), stitched later on the device (``stack.process_stream(overlap=...)``)
    -- of the files' dtype: a NumPy array, or with ``pinned`` a page-locked torch tensor ready for an
    asynchronous upload.  One assay per pattern."""
    path_dict, _ = extract_paths(os.fspath(pattern), assay="str", channel="str", time="time", row="int", col="int")
    if len(path_dict) == 0:
        raise FileNotFoundError(f"The pattern {pattern} did not lead to any files.")
    if len({k[0] for k in path_dict}) > 1:
        raise ValueError("iter_time_chunks streams one assay per pattern")
    in_file, inner, (h, w), dtype, _ = _tiff_layout(next(iter(path_dict.values())))
    path_dims = {"channel": any(k[1] is not None for k in path_dict), "time": any(k[2] is not None for k in path_dict)}
    for d in in_file:
        if path_dims[d]:
            raise ValueError("Dimensions specified in the path names and inside the tiff file overlap.")
    if not path_dims["time"] and "time" not in in_file:
        raise ValueError("the pattern needs a (time) group, or files with a time axis")
    tiled = any(k[3] is not None or k[4] is not None for k in path_dict)
    rows = sorted({k[3] for k in path_dict}, key=lambda v: (v is None, v))
    cols = sorted({k[4] for k in path_dict}, key=lambda v: (v is None, v))
    n_t_file = inner[in_file.index("time")] if "time" in in_file else 1
    n_c_file = inner[in_file.index("channel")] if "channel" in in_file else 1
    channels = sorted({k[1] for k in path_dict}, key=lambda c: (c is None, c)) if path_dims["channel"] else list(range(n_c_file))
    times = sorted({k[2] for k in path_dict}) if path_dims["time"] else list(range(n_t_file))

    def page_of(t, c, r, cc):  # -> (path, page index inside the file): pages run (time, channel), channel fastest
        key = (next(iter(path_dict))[0], c if path_dims["channel"] else None, t if path_dims["time"] else None, r, cc)
        if key not in path_dict:
            raise FileNotFoundError(f"no file for channel {c!r}, time {t}, tile ({r}, {cc})")
        ti = 0 if path_dims["time"] else t
        ci = 0 if path_dims["channel"] else c
        return path_dict[key], ti * n_c_file + ci

    for lo in range(0, len(times), int(chunk)):
        part = times[lo: lo + int(chunk)]
        shape = (len(part), len(channels)) + ((len(rows), len(cols)) if tiled else ()) + (h, w)
        if pinned:
            import torch

            block_t = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype).pin_memory()
            block = block_t.numpy()
        else:
            block = np.empty(shape, dtype=dtype)
        for i, t in enumerate(part):
            for j, c in enumerate(channels):
                for a, r in enumerate(rows):
                    for b, cc in enumerate(cols):
                        path, index = page_of(t, c, r, cc)
                        dst = block[i, j, a, b] if tiled else block[i, j]
                        _read_page(path, index, dst)
        stamps = [int(t.timestamp()) if isinstance(t, datetime.datetime) else t for t in part]
        yield stamps, [c for c in channels], (block_t if pinned else block)
