"""Reader (reference: src/magnify/reader.py).

In-memory inputs (DataArray / Dataset objects, magnify_amd's or real xarray's, or a sequence of
them) pass straight through (reader.py:31-35).  Path patterns follow the reference's grammar
(reader.py:80-160): named groups ``(assay)``, ``(channel)``, ``(time)`` / ``(time|%Y%m%d)``,
``(row)``, ``(col)`` select the file's place in the tile array, ``(name_key|formatter|format)`` groups
attach an alternative labelling ``name`` to dimension ``key``; ``*`` / ``?`` / ``**`` glob as usual.
Files are TIFFs read with this build's own TIFF layer (``magnify_amd.tiff``: classic and BigTIFF, strips or tiles,
the usual lossless codecs): one 2-D page per file, or a series whose axes the file itself names -- OME-XML
(``DimensionOrder`` / ``SizeC`` / ``SizeT``, ``Plane DeltaT``), MicroManager's summary (``StartTime``, ``ChNames``) or
an ImageJ hyperstack -- with the reference's rules (reader.py:194-258): a Z axis is refused, the position axis ``R``
is ignored (tiles come from the path), X and Y must be there, a dimension named both in the path and in the file is
an error.  ``read_tiffs`` fills one host array per assay page by page; SURVEY 8f N2's streaming half is
``iter_time_chunks``: the series chunk by chunk of the time axis into page-locked blocks, one page decoded at a time
(the reference maps every page to its own dask block, reader.py:265-292), feeding ``stack.process_stream``.
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


_LETTER_TO_DIM = {"C": "channel", "T": "time", "Z": "depth", "Y": "tile_y", "X": "tile_x", "R": "tile_pos"}  # reader.py:196-203


def series_layout(path, need_times=True, need_channels=True):
    """What the reference learns from the first file of an assay (reader.py:189-258): the dimensions inside the file
    (position axis dropped), the shape of the axes in front of a page, page size and dtype, and -- when the path does
    not give them -- the timepoints (MicroManager ``StartTime`` + OME ``Plane DeltaT``) and channel names (``ChNames``).
    -> dict(dims, inner, page, dtype, times, channels)."""
    from . import tiff

    with tiff.TiffFile(path) as tif:
        axes, shape = tif.series()
        unknown = [c for c in axes if c not in _LETTER_TO_DIM]
        if unknown:
            raise ValueError(f"{path}: {shape[0]} pages with no description of their axes (tifffile axes {axes!r}); "
                             "name the dimension in the path or write OME / ImageJ metadata")
        dims = [_LETTER_TO_DIM[c] for c in axes]
        inner = tuple(shape)
        times = channels = None
        summary = tif.micromanager_metadata.get("Summary", {}) if tif.is_micromanager else {}
        if need_times and "StartTime" in summary:
            start = datetime.datetime.strptime(summary["StartTime"][:-6], "%Y-%m-%d %H:%M:%S.%f")  # without the time zone
            if "time" in dims:
                planes = tif.ome_planes or []
                if not all(pl.get("DeltaTUnit") == "ms" for pl in planes):
                    raise ValueError(f"{path}: OME Plane DeltaT in a unit other than ms")
                times = [start + datetime.timedelta(milliseconds=float(pl.get("DeltaT"))) for pl in planes]
                stride = inner[dims.index("channel")] if "channel" in dims else 1
                if len(times) % stride:
                    raise ValueError(f"{path}: {len(times)} OME planes for {stride} channels")
                times = times[::stride]
            else:
                times = [start]
        if need_channels and "ChNames" in summary:
            channels = list(summary["ChNames"])
        if "tile_pos" in dims:  # positions are separate files: the user names tiles in the search path (reader.py:249-254)
            k = dims.index("tile_pos")
            inner, dims = inner[:k] + inner[k + 1:], dims[:k] + dims[k + 1:]
        if "depth" in dims:
            raise ValueError("tiff files with a Z dimension are not yet supported.")
        if "tile_y" not in dims or "tile_x" not in dims:
            raise ValueError("tiff files must contain an X and Y dimension.")
        page = tif.page(0)
        return {"dims": dims[:-2], "inner": tuple(inner[:-2]), "page": page.shape, "dtype": tif.dtype, "times": times,
                "channels": channels}


def read_tiffs(xp_dict, name, meta_dict):
    """reader.py:163-324: one Dataset with ``tile`` over the dimensions found in the paths and inside
    the files, in the standard order (channel, time, tile_row, tile_col, tile_y, tile_x)."""
    from . import tiff

    channel_idx, time_idx, row_idx, col_idx = (sorted(set(i)) for i in zip(*xp_dict.keys()))
    in_path, outer = [], ()
    for dim, values in (("channel", channel_idx), ("time", time_idx), ("tile_row", row_idx), ("tile_col", col_idx)):
        if values[0] != -1:
            in_path.append(dim)
            outer += (len(values),)
    files = [p for _, p in sorted(xp_dict.items())]
    lay = series_layout(files[0], need_times="time" not in in_path, need_channels="channel" not in in_path)
    in_file, inner, (ty, tx) = lay["dims"], lay["inner"], lay["page"]
    if set(in_file) & set(in_path):
        raise ValueError("Dimensions specified in the path names and inside the tiff file overlap.")
    if len(files) != int(np.prod(outer, dtype=np.int64)):
        raise ValueError(f"{name or 'assay'}: {len(files)} files do not fill the {outer} array the pattern describes")
    tiles = np.empty(outer + inner + (ty, tx), dtype=lay["dtype"])
    flat = tiles.reshape((-1,) + (ty, tx))
    per_file = int(np.prod(inner, dtype=np.int64)) if inner else 1
    for f, path in enumerate(files):
        with tiff.TiffFile(path) as tif:
            for k in range(per_file):  # pages run over the in-file axes in C order (reader.py:272-282)
                tif.read_page_into(k, flat[f * per_file + k])
    times = time_idx if "time" in in_path else lay["times"]
    channels = channel_idx if "channel" in in_path else lay["channels"]
    dims = tuple(in_path + in_file + ["tile_y", "tile_x"])
    coords = {}
    if channels is not None and ("channel" not in dims or len(channels) == tiles.shape[dims.index("channel")]):
        coords["channel"] = list(channels)
    if times is not None and ("time" not in dims or len(times) == tiles.shape[dims.index("time")]):
        coords["time"] = [int(t.timestamp()) if isinstance(t, datetime.datetime) else t for t in times]  # seconds
    xp = xr_lite.Dataset({"tile": xr_lite.DataArray(tiles, dims)}, coords=coords, attrs={"name": name})
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


class _BlockPool:
    """Free host blocks by (shape, dtype, pinned), kept by the module so that a second pass over a series pins nothing
    new (pinning a fresh 1.3 GB block per chunk cost 0.2 s, six times the reading).  A block belongs to ONE live
    iterator at a time: ``take`` removes it from the pool, the iterator gives it back when it ends."""

    def __init__(self):
        import threading

        self._free, self._lock = collections.defaultdict(list), threading.Lock()

    def take(self, shape, dtype, pinned):
        key = (tuple(shape), str(dtype), bool(pinned))
        with self._lock:
            if self._free[key]:
                return self._free[key].pop()
        if pinned:
            import torch

            return torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype).pin_memory()
        return np.empty(shape, dtype=dtype)

    def give(self, block, pinned):
        key = (tuple(block.shape), str(block.numpy().dtype if pinned else block.dtype), bool(pinned))
        with self._lock:
            self._free[key].append(block)

    def clear(self):
        with self._lock:
            self._free.clear()


_POOL = _BlockPool()


def release_pinned():
    """Drop the host blocks ``iter_time_chunks`` keeps for the next series (page-locked ones included)."""
    _POOL.clear()


class TimeSeries:
    """What ``iter_time_chunks`` learns from a pattern before it reads a pixel: the files, the axes inside them, the
    timepoints and channels.  One assay per pattern.  ``len(series)`` = timepoints; ``series.chunks(...)`` streams
    them (any contiguous range of them: the reference's unit is one dask block per TIFF page, reader.py:265-292, and
    one assay per loop turn, pipeline.py:18-24, so any consumer -- a rank of a multi-GPU run -- can take any time
    range)."""

    def __init__(self, pattern):
        self.pattern = os.fspath(pattern)
        path_dict, _ = extract_paths(self.pattern, assay="str", channel="str", time="time", row="int", col="int")
        if len(path_dict) == 0:
            raise FileNotFoundError(f"The pattern {pattern} did not lead to any files.")
        if len({k[0] for k in path_dict}) > 1:
            raise ValueError("iter_time_chunks streams one assay per pattern")
        self.path_dict = path_dict
        self.path_dims = {"channel": any(k[1] is not None for k in path_dict), "time": any(k[2] is not None for k in path_dict)}
        lay = series_layout(next(iter(path_dict.values())), need_times=not self.path_dims["time"],
                            need_channels=not self.path_dims["channel"])
        self.in_file, self.inner, self.page_shape, self.dtype = lay["dims"], lay["inner"], lay["page"], lay["dtype"]
        for d in self.in_file:
            if self.path_dims[d]:
                raise ValueError("Dimensions specified in the path names and inside the tiff file overlap.")
        if not self.path_dims["time"] and "time" not in self.in_file:
            raise ValueError("the pattern needs a (time) group, or files with a time axis")
        self.tiled = any(k[3] is not None or k[4] is not None for k in path_dict)
        self.rows = sorted({k[3] for k in path_dict}, key=lambda v: (v is None, v))
        self.cols = sorted({k[4] for k in path_dict}, key=lambda v: (v is None, v))
        in_file, inner = self.in_file, self.inner
        n_t_file = inner[in_file.index("time")] if "time" in in_file else 1
        n_c_file = inner[in_file.index("channel")] if "channel" in in_file else 1
        # page index inside a file: C order over the in-file axes, whichever of time / channel comes first
        self.stride = {d: int(np.prod(inner[in_file.index(d) + 1:], dtype=np.int64)) for d in in_file}
        self.channels = (sorted({k[1] for k in path_dict}, key=lambda c: (c is None, c)) if self.path_dims["channel"]
                         else list(range(n_c_file)))
        self.times = sorted({k[2] for k in path_dict}) if self.path_dims["time"] else list(range(n_t_file))
        use_file_times = not (self.path_dims["time"] or lay["times"] is None or len(lay["times"]) != n_t_file)
        self.time_values = lay["times"] if use_file_times else self.times
        use_file_names = not (self.path_dims["channel"] or lay["channels"] is None or len(lay["channels"]) != n_c_file)
        self.channel_names = list(lay["channels"] if use_file_names else self.channels)
        self.assay = next(iter(path_dict))[0]

    def __len__(self):
        return len(self.times)

    def block_shape(self, n_t):
        return (n_t, len(self.channels)) + ((len(self.rows), len(self.cols)) if self.tiled else ()) + tuple(self.page_shape)

    def page_of(self, t, c, r, cc):
        """-> (path, page index inside the file) of timepoint value ``t``, channel value ``c``, tile (r, cc)."""
        key = (self.assay, c if self.path_dims["channel"] else None, t if self.path_dims["time"] else None, r, cc)
        if key not in self.path_dict:
            raise FileNotFoundError(f"no file for channel {c!r}, time {t}, tile ({r}, {cc})")
        index = ((0 if self.path_dims["time"] else t * self.stride.get("time", 0))
                 + (0 if self.path_dims["channel"] else c * self.stride.get("channel", 0)))
        return self.path_dict[key], index

    def chunks(self, chunk, time_range=None, pinned=False, workers=None, ring=None):
        return _ChunkIter(self, chunk, time_range, pinned, workers, ring)


class _ChunkIter:
    """Iterator behind ``iter_time_chunks``.  ``.first_timepoint`` = the global index of the first timepoint it yields
    (what ``stack.process_stream(first_timepoint=...)`` needs to number assays and seeds as the unsharded run does),
    ``.ring`` = how many blocks it cycles through (None: a fresh block per chunk)."""

    def __init__(self, series, chunk, time_range, pinned, workers, ring):
        self.series, self.chunk, self.pinned = series, int(chunk), bool(pinned)
        if self.chunk < 1:
            raise ValueError("chunk must be at least one timepoint")
        lo, hi = (0, len(series)) if time_range is None else (int(time_range[0]), int(time_range[1]))
        if not 0 <= lo <= hi <= len(series):
            raise ValueError(f"time_range {time_range} outside the series' {len(series)} timepoints")
        self.lo, self.hi, self.first_timepoint = lo, hi, lo
        if ring is None:
            ring = 4 if pinned else None  # a fresh pageable block per chunk unless asked otherwise
        if ring is not None and int(ring) < 1:
            raise ValueError("ring must be at least 1")
        self.ring = None if ring is None else int(ring)
        if workers is None:
            workers = min(16, os.cpu_count() or 1)
        self.workers = max(1, int(workers))
        self._gen = self._run()

    def __iter__(self):
        return self

    def __next__(self):
        return next(self._gen)

    def close(self):
        self._gen.close()

    def _run(self):
        from . import tiff

        ser = self.series
        open_files, blocks = {}, []  # path -> TiffFile, kept open across chunks; the blocks this iterator holds

        def opened(path):
            tif = open_files.get(path)
            if tif is None:
                if len(open_files) >= 256:
                    open_files.pop(next(iter(open_files))).close()
                tif = open_files[path] = tiff.TiffFile(path)
            return tif

        try:
            n_yielded = 0
            for lo in range(self.lo, self.hi, self.chunk):
                part = ser.times[lo: min(lo + self.chunk, self.hi)]
                shape = ser.block_shape(len(part))
                if self.ring is None:
                    block_t = np.empty(shape, dtype=ser.dtype)
                else:
                    slot = n_yielded % self.ring
                    if slot < len(blocks) and tuple(blocks[slot].shape) != shape:  # the shorter last chunk
                        _POOL.give(blocks[slot], self.pinned)
                        blocks[slot] = _POOL.take(shape, ser.dtype, self.pinned)
                    elif slot >= len(blocks):
                        blocks.append(_POOL.take(shape, ser.dtype, self.pinned))
                    block_t = blocks[slot]
                block = block_t.numpy() if self.pinned else block_t
                n_yielded += 1
                pages = []  # (TiffFile, page index, destination)
                for i, t in enumerate(part):
                    for j, c in enumerate(ser.channels):
                        for a, r in enumerate(ser.rows):
                            for b, cc in enumerate(ser.cols):
                                path, index = ser.page_of(t, c, r, cc)
                                pages.append((opened(path), index, block[i, j, a, b] if ser.tiled else block[i, j]))
                tiff.read_pages(pages, self.workers)
                stamps = [int(t.timestamp()) if isinstance(t, datetime.datetime) else t
                          for t in ser.time_values[lo: lo + len(part)]]
                yield stamps, list(ser.channel_names), block_t
        finally:
            for tif in open_files.values():
                tif.close()
            for blk in blocks:
                _POOL.give(blk, self.pinned)


def iter_time_chunks(pattern, chunk: int, pinned: bool = False, workers: int | None = None, ring: int | None = None,
                     time_range=None, rank: int | None = None, world: int | None = None):
    """Streamed ingest of a time series too large to hold: the files behind
    ``pattern`` are read ``chunk`` timepoints at a time, in time order, page by page, never all at once.
    Groups as in ``extract_paths``: ``(channel)``, ``(time|format)``, and for tiled acquisitions ``(row)`` /
    ``(col)``; a file holds one 2-D page or a series (OME-TIFF / BigTIFF, MicroManager, ImageJ hyperstack) whose pages
    run over time and / or channel (a dimension is either in the path or in the file, reader.py:260-262); timepoints
    and channel names come from the path or, failing that, from the file's metadata (``series_layout``).
    Yields ``(time_values, channels, block)`` with ``block`` (T_chunk, C, H, W) -- tiled series:
    (T_chunk, C, rows, cols, tile_y, tile_x), stitched later on the device (``stack.process_stream(overlap=...)``)
    -- of the files' dtype: a NumPy array, or with ``pinned`` a page-locked torch tensor ready for an
    asynchronous upload.

    ``time_range=(lo, hi)`` (or ``rank`` / ``world``: this rank's ``distributed.shard_range`` block) restricts the
    stream to the timepoints [lo, hi) of the series -- config C5 across GPUs: every rank streams its own contiguous
    block of the time axis; the returned iterator's ``first_timepoint`` (= lo) goes to
    ``stack.process_stream(first_timepoint=...)`` so that assay indices, seeds and file names are those of the
    unsharded run.

    A chunk's pages are read by ``workers`` threads side by side, outside the interpreter (``mg_host_read_runs``:
    positional reads straight into the block) when the pages are stored uncompressed in one piece -- what acquisition
    software writes --, page by page through ``TiffFile.read_page_into`` otherwise.  Page-locked blocks come from a
    ring of ``ring`` buffers (default 4) that belongs to THIS iterator: a yielded block is overwritten once ``ring - 1``
    further chunks have been yielded -- ``stack.process_stream`` holds at most ``prefetch + 2`` and checks --; without
    ``pinned`` every chunk is a fresh array unless ``ring`` is given.  One assay per pattern."""
    series = pattern if isinstance(pattern, TimeSeries) else TimeSeries(pattern)
    if rank is not None or world is not None:
        if time_range is not None:
            raise ValueError("iter_time_chunks: time_range or rank / world, not both")
        from .distributed import shard_range

        time_range = shard_range(len(series), int(rank or 0), int(world or 1))
    return series.chunks(chunk, time_range=time_range, pinned=pinned, workers=workers, ring=ring)
