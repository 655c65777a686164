"""The TIFF layer the streamed ingest needs (reference: src/magnify/reader.py:194-292, which leans on tifffile).

A small IFD walker for classic TIFF (magic 42) and BigTIFF (magic 43, what a terabyte-scale acquisition is written
as), either byte order: per page the geometry, sample type and the strip / tile offsets; pages are decoded one at a
time, straight from the file into a caller's buffer (`read_page_into`: one `readinto` for an uncompressed page whose
strips lie back to back -- the common case for acquisition software), so a series is never held whole.  Codecs:
none, Deflate (8 / 32946), PackBits (32773), LZW (5); horizontal predictor 2.  Grayscale pages only.

The series axes come, in tifffile's order of preference, from
  * OME-XML in the first page's ImageDescription (``Pixels DimensionOrder / SizeC / SizeT / SizeZ``; ``Plane DeltaT``;
    ``Channel Name``); several ``Image`` elements (MicroManager positions) make a leading ``R`` axis when this file
    holds them all, else the file's own positions only,
  * an ImageJ hyperstack description (``images / channels / slices / frames``),
  * else one page = ``YX``; several undescribed pages = ``IYX`` (a bare sequence: the reference has no dimension for
    it and fails, reader.py:196-204).
Singleton axes are dropped, as tifffile's squeezed ``series[0].axes`` does.  MicroManager's summary block (the
``StartTime`` and ``ChNames`` the reference reads, reader.py:211-247) sits behind the classic header at byte 8.
"""
from __future__ import annotations

import json
import os
import threading
import re
import struct
import zlib
from xml.etree import ElementTree

import numpy as np

_TYPE_FMT = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 7: "B", 8: "h", 9: "i", 10: "ii", 11: "f", 12: "d",
             13: "I", 16: "Q", 17: "q", 18: "Q"}
_TYPE_SIZE = {1: 1, 2: 1, 3: 2, 4: 4, 5: 8, 6: 1, 7: 1, 8: 2, 9: 4, 10: 8, 11: 4, 12: 8, 13: 4, 16: 8, 17: 8, 18: 8}

T_WIDTH, T_LENGTH, T_BITS, T_COMPRESSION, T_PHOTOMETRIC, T_DESCRIPTION = 256, 257, 258, 259, 262, 270
T_STRIP_OFFSETS, T_SAMPLES, T_ROWS_PER_STRIP, T_STRIP_COUNTS, T_PLANAR, T_SOFTWARE = 273, 277, 278, 279, 284, 305
T_PREDICTOR, T_TILE_WIDTH, T_TILE_LENGTH, T_TILE_OFFSETS, T_TILE_COUNTS, T_SAMPLE_FORMAT = 317, 322, 323, 324, 325, 339
T_MM_METADATA = 51123  # MicroManagerMetadata (per-page JSON); its presence is tifffile's `is_micromanager`


class TiffError(ValueError):
    pass


class TiffPage:
    """One IFD: geometry, dtype and where its strips / tiles lie."""

    __slots__ = ("index", "width", "length", "dtype", "compression", "predictor", "offsets", "counts", "rows_per_strip",
                 "tile", "description", "tags", "samples")

    @property
    def shape(self):
        return (self.length, self.width)

    @property
    def contiguous(self):
        """(offset, nbytes) when the page is uncompressed and its strips lie back to back in row order, else None."""
        if self.compression != 1 or self.tile is not None or self.predictor != 1:
            return None
        pos = self.offsets[0]
        for off, cnt in zip(self.offsets, self.counts):
            if off != pos:
                return None
            pos += cnt
        nbytes = self.length * self.width * self.dtype.itemsize
        return (self.offsets[0], nbytes) if pos - self.offsets[0] >= nbytes else None


class TiffFile:
    """``with TiffFile(path) as tif``: ``tif.pages`` (parsed lazily, IFD by IFD), ``tif.axes`` / ``tif.shape`` of the
    first series, ``tif.read_page_into(i, out)``."""

    def __init__(self, path):
        self.path = os.fspath(path)
        self._fh = open(self.path, "rb")
        self._size = os.fstat(self._fh.fileno()).st_size
        head = self._fh.read(16)
        if len(head) < 8 or head[:2] not in (b"II", b"MM"):
            raise TiffError(f"{self.path}: not a TIFF file")
        self.bo = "<" if head[:2] == b"II" else ">"
        magic = struct.unpack(self.bo + "H", head[2:4])[0]
        if magic == 42:
            self.big = False
            self._next = struct.unpack(self.bo + "I", head[4:8])[0]
        elif magic == 43:
            self.big = True
            size, zero = struct.unpack(self.bo + "HH", head[4:8])
            if size != 8 or zero != 0:
                raise TiffError(f"{self.path}: malformed BigTIFF header")
            self._next = struct.unpack(self.bo + "Q", head[8:16])[0]
        else:
            raise TiffError(f"{self.path}: not a TIFF file (magic {magic})")
        self._pages = []
        self._series = None
        self._mm = None
        self._lock = threading.Lock()  # the IFD walk; pixel reads are positional (pread) and need none

    # -- context management ---------------------------------------------------------------------------------
    def close(self):
        if self._fh is not None:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- IFD walk -------------------------------------------------------------------------------------------------
    def _read_at(self, offset, n):
        if offset < 0 or offset + n > self._size:
            raise TiffError(f"{self.path}: read of {n} bytes at {offset} beyond the end of the file ({self._size})")
        data = os.pread(self._fh.fileno(), n, offset)  # positional: several threads may read pages of one file
        while len(data) < n:
            more = os.pread(self._fh.fileno(), n - len(data), offset + len(data))
            if not more:
                raise TiffError(f"{self.path}: short read at {offset}")
            data += more
        return data

    def _values(self, typ, count, raw_value):
        """Decode a tag's values; `raw_value` = the 4 / 8 inline bytes (value or offset)."""
        if typ not in _TYPE_SIZE:
            return None
        nbytes = _TYPE_SIZE[typ] * count
        inline = 8 if self.big else 4
        if nbytes <= inline:
            data = raw_value[:nbytes]
        else:
            off = struct.unpack(self.bo + ("Q" if self.big else "I"), raw_value)[0]
            data = self._read_at(off, nbytes)
        if typ == 2:
            return data.split(b"\x00", 1)[0].decode("utf-8", "replace") if count else ""
        if typ == 7:
            return data
        fmt = _TYPE_FMT[typ]
        if len(fmt) == 2:  # rationals: pairs
            vals = struct.unpack(self.bo + fmt[0] * (2 * count), data)
            return tuple((vals[2 * i], vals[2 * i + 1]) for i in range(count))
        return struct.unpack(self.bo + fmt * count, data)

    def _parse_next(self):
        """Parse the IFD at self._next into a TiffPage; returns False at the end of the chain."""
        if not self._next:
            return False
        off = self._next
        if self.big:
            n = struct.unpack(self.bo + "Q", self._read_at(off, 8))[0]
            entry, body = 20, self._read_at(off + 8, n * 20 + 8)
            self._next = struct.unpack(self.bo + "Q", body[n * 20:])[0]
        else:
            n = struct.unpack(self.bo + "H", self._read_at(off, 2))[0]
            entry, body = 12, self._read_at(off + 2, n * 12 + 4)
            self._next = struct.unpack(self.bo + "I", body[n * 12:])[0]
        if self._next == off:
            raise TiffError(f"{self.path}: IFD chain loops")
        tags = {}
        for i in range(n):
            e = body[i * entry: (i + 1) * entry]
            if self.big:
                tag, typ, count = struct.unpack(self.bo + "HHQ", e[:12])
                tags[tag] = (typ, count, e[12:20])
            else:
                tag, typ, count = struct.unpack(self.bo + "HHI", e[:8])
                tags[tag] = (typ, count, e[8:12])

        def get(tag, default=None):
            if tag not in tags:
                return default
            v = self._values(*tags[tag])
            return default if v is None else v

        page = TiffPage()
        page.index = len(self._pages)
        page.tags = tags
        page.width, page.length = int(get(T_WIDTH, (0,))[0]), int(get(T_LENGTH, (0,))[0])
        if page.width <= 0 or page.length <= 0:
            raise TiffError(f"{self.path}: page {page.index} has no size")
        page.samples = int(get(T_SAMPLES, (1,))[0])
        bits = get(T_BITS, (1,))
        fmt = int(get(T_SAMPLE_FORMAT, (1,))[0])
        if page.samples != 1 or len(set(bits)) != 1:
            raise TiffError(f"{self.path}: page {page.index} has {page.samples} samples per pixel; grayscale only")
        kind = {1: "u", 2: "i", 3: "f", 4: "u"}.get(fmt)
        if kind is None or bits[0] not in (8, 16, 32, 64) or (kind == "f" and bits[0] < 32):
            raise TiffError(f"{self.path}: page {page.index}: {bits[0]}-bit sample format {fmt} is not supported")
        page.dtype = np.dtype(f"{self.bo}{kind}{bits[0] // 8}")
        page.compression = int(get(T_COMPRESSION, (1,))[0])
        page.predictor = int(get(T_PREDICTOR, (1,))[0])
        page.description = get(T_DESCRIPTION, None)
        if T_TILE_OFFSETS in tags:
            page.tile = (int(get(T_TILE_LENGTH)[0]), int(get(T_TILE_WIDTH)[0]))
            page.offsets, page.counts = tuple(get(T_TILE_OFFSETS)), tuple(get(T_TILE_COUNTS))
            page.rows_per_strip = page.tile[0]
        else:
            page.tile = None
            page.offsets = tuple(get(T_STRIP_OFFSETS, ()))
            page.counts = tuple(get(T_STRIP_COUNTS, ()))
            page.rows_per_strip = min(int(get(T_ROWS_PER_STRIP, (page.length,))[0]), page.length)
            if not page.offsets:
                raise TiffError(f"{self.path}: page {page.index} has no strips")
            if not page.counts and len(page.offsets) == 1:  # (old writers leave the count out of one-strip pages)
                page.counts = (page.length * page.width * page.dtype.itemsize,)
        if len(page.offsets) != len(page.counts):
            raise TiffError(f"{self.path}: page {page.index}: offsets and byte counts differ in number")
        self._pages.append(page)
        return True

    def page(self, index):
        if len(self._pages) <= index:
            with self._lock:
                while len(self._pages) <= index:
                    if not self._parse_next():
                        raise IndexError(f"{self.path}: page {index} of {len(self._pages)}")
        return self._pages[index]

    @property
    def pages(self):
        with self._lock:
            while self._parse_next():
                pass
        return self._pages

    def __len__(self):
        return len(self.pages)

    # -- metadata -----------------------------------------------------------------------------------------------
    @property
    def is_micromanager(self):
        return T_MM_METADATA in self.page(0).tags

    @property
    def micromanager_metadata(self):
        """{'Summary': {...}} from MicroManager's header block (classic TIFF: eight uint32 behind the 8-byte header,
        the fourth pair being (2355492, length) in front of the summary JSON)."""
        if self._mm is None:
            self._mm = {}
            try:
                vals = struct.unpack(self.bo + "8I", self._read_at(8, 32))
                if vals[6] == 2355492 and 0 < vals[7] <= self._size - 40:
                    self._mm["Summary"] = json.loads(self._read_at(40, vals[7]).decode("utf-8", "replace").rstrip("\x00"))
            except (TiffError, ValueError, struct.error):
                self._mm = {}
        return self._mm

    def _ome(self):
        desc = self.page(0).description
        if not desc or "<OME" not in desc[:4096] or not desc.lstrip().startswith("<"):
            return None
        try:
            root = ElementTree.fromstring(desc.encode("utf-8"))
        except ElementTree.ParseError as exc:
            raise TiffError(f"{self.path}: unreadable OME-XML: {exc}") from None
        strip = lambda t: t.rsplit("}", 1)[-1]  # noqa: E731  (namespaces vary with the schema year)
        images = [e for e in root.iter() if strip(e.tag) == "Image"]
        out = []
        for im in images:
            px = next((e for e in im if strip(e.tag) == "Pixels"), None)
            if px is None:
                continue
            planes = [dict(e.attrib) for e in px if strip(e.tag) == "Plane"]
            chans = [e.attrib.get("Name") for e in px if strip(e.tag) == "Channel"]
            data = [dict(e.attrib, **{"FileName": next((u.attrib.get("FileName") for u in e if strip(u.tag) == "UUID"), None)})
                    for e in px if strip(e.tag) == "TiffData"]
            out.append({"order": px.attrib.get("DimensionOrder", "XYCZT"),
                        "sizes": {a: int(px.attrib.get("Size" + a, 1)) for a in "XYCZT"},
                        "planes": planes, "channels": chans, "tiffdata": data, "name": im.attrib.get("Name")})
        return out or None

    @property
    def ome_planes(self):
        """``Plane`` attribute dicts of the first OME ``Image`` (DeltaT, DeltaTUnit, TheC, TheT, TheZ), or None."""
        ome = self._ome()
        return ome[0]["planes"] if ome else None

    @property
    def ome_channel_names(self):
        ome = self._ome()
        return ome[0]["channels"] if ome else None

    def series(self):
        """(axes, shape) of the first series, singleton axes dropped: letters of ``CTZYXRI`` as tifffile names them."""
        if self._series is not None:
            return self._series
        p0 = self.page(0)
        ome = self._ome()
        n_pages = None
        if ome:
            im = ome[0]
            order = im["order"]
            if sorted(order) != sorted("XYCZT") or order[:2] != "XY":
                raise TiffError(f"{self.path}: OME DimensionOrder {order!r}")
            sizes = im["sizes"]
            mine = os.path.basename(self.path)
            # positions (several Image elements): an R axis when this file holds them all, else one position per file
            here = [i for i in ome if not i["tiffdata"] or any(d.get("FileName") in (None, mine) for d in i["tiffdata"])]
            n_pos = len(here) if len(here) > 1 and all(i["sizes"] == sizes and i["order"] == order for i in here) else 1
            axes = ("R" if n_pos > 1 else "") + order[:1:-1] + "YX"       # slowest first: reversed(order[2:]) then Y, X
            shape = ((n_pos,) if n_pos > 1 else ()) + tuple(sizes[a] for a in order[:1:-1]) + (sizes["Y"], sizes["X"])
        else:
            desc = p0.description or ""
            if desc.startswith("ImageJ="):
                found = {k: int(v) for k, v in re.findall(r"(images|channels|slices|frames)=(\d+)", desc)}
                n_c, n_z, n_t = found.get("channels", 1), found.get("slices", 1), found.get("frames", 1)
                n_pages = found.get("images", n_c * n_z * n_t)
                if n_c * n_z * n_t != n_pages:  # (ImageJ writes `images` alone for a plain stack)
                    axes, shape = "IYX", (n_pages, p0.length, p0.width)
                else:
                    axes, shape = "TZCYX", (n_t, n_z, n_c, p0.length, p0.width)
            else:
                n_pages = len(self.pages)
                axes, shape = ("IYX", (n_pages, p0.length, p0.width)) if n_pages > 1 else ("YX", (p0.length, p0.width))
        if (p0.length, p0.width) != shape[-2:]:
            raise TiffError(f"{self.path}: the description's image size {shape[-2:]} is not the page size {(p0.length, p0.width)}")
        keep = [i for i, n in enumerate(shape) if n != 1 or i >= len(shape) - 2]
        self._series = ("".join(axes[i] for i in keep), tuple(shape[i] for i in keep))
        return self._series

    @property
    def axes(self):
        return self.series()[0]

    @property
    def shape(self):
        return self.series()[1]

    @property
    def dtype(self):
        return self.page(0).dtype.newbyteorder("=")

    # -- pixel data -----------------------------------------------------------------------------------------------
    def read_page_into(self, index, out):
        """Decode page ``index`` into ``out`` (a writable C-contiguous (H, W) array of the page's native dtype)."""
        page = self.page(index)
        if out.shape != page.shape or out.dtype != page.dtype.newbyteorder("=") or not out.flags.c_contiguous:
            raise TiffError(f"{self.path}: page {index} is {page.shape} {page.dtype}, the buffer {out.shape} {out.dtype}")
        swap = page.dtype.byteorder not in ("=", "|") and page.dtype != page.dtype.newbyteorder("=")
        run = page.contiguous
        if run is not None:
            off, nbytes = run
            if off + nbytes > self._size:
                raise TiffError(f"{self.path}: page {index} reaches beyond the end of the file")
            view, got = memoryview(out).cast("B"), 0
            while got < nbytes:  # straight into the caller's (page-locked) buffer, no seek: thread-safe
                n = os.preadv(self._fh.fileno(), [view[got:]], off + got)
                if n <= 0:
                    raise TiffError(f"{self.path}: short read in page {index}")
                got += n
            if swap:
                out.byteswap(inplace=True)
            return out
        item = page.dtype.itemsize
        if page.tile is None:
            rps = page.rows_per_strip
            for s, (off, cnt) in enumerate(zip(page.offsets, page.counts)):
                r0 = s * rps
                rows = min(rps, page.length - r0)
                if rows <= 0:
                    break
                seg = _decode(self._read_at(off, cnt), page.compression, rows * page.width * item, self.path)
                block = np.frombuffer(seg, dtype=page.dtype, count=rows * page.width).reshape(rows, page.width)
                out[r0: r0 + rows] = _unpredict(block, page.predictor)
        else:
            th, tw = page.tile
            across = (page.width + tw - 1) // tw
            for t, (off, cnt) in enumerate(zip(page.offsets, page.counts)):
                r0, c0 = (t // across) * th, (t % across) * tw
                if r0 >= page.length:
                    break
                seg = _decode(self._read_at(off, cnt), page.compression, th * tw * item, self.path)
                block = _unpredict(np.frombuffer(seg, dtype=page.dtype, count=th * tw).reshape(th, tw), page.predictor)
                rows, cols = min(th, page.length - r0), min(tw, page.width - c0)
                out[r0: r0 + rows, c0: c0 + cols] = block[:rows, :cols]
        return out

    def page_run(self, index, out):
        """(offset, nbytes) if page ``index`` can be read into ``out`` by ONE positional read (uncompressed, strips back
        to back, native byte order), else None.  Checks ``out`` like ``read_page_into``."""
        page = self.page(index)
        if out.shape != page.shape or out.dtype != page.dtype.newbyteorder("=") or not out.flags.c_contiguous:
            raise TiffError(f"{self.path}: page {index} is {page.shape} {page.dtype}, the buffer {out.shape} {out.dtype}")
        if page.dtype.byteorder not in ("=", "|") and page.dtype != page.dtype.newbyteorder("="):
            return None
        run = page.contiguous
        if run is not None and run[0] + run[1] > self._size:
            raise TiffError(f"{self.path}: page {index} reaches beyond the end of the file")
        return run

    def fileno(self):
        return self._fh.fileno()

    def asarray(self, index=0):
        page = self.page(index)
        return self.read_page_into(index, np.empty(page.shape, dtype=page.dtype.newbyteorder("=")))


def read_pages(pages, workers=1):
    """Decode many pages side by side: ``pages`` = [(TiffFile, page index, out array)].  Pages that are one byte run
    in their file (``TiffFile.page_run``) are read by ``mg_host_read_runs`` -- ``workers`` native threads, positional
    reads straight into the destinations, no interpreter between two pages --; the others (compressed, tiled, foreign
    byte order) go page by page through ``read_page_into`` on a thread pool."""
    runs, slow = [], []
    for tif, index, out in pages:
        run = tif.page_run(index, out)
        if run is None:
            slow.append((tif, index, out))
        else:
            runs.append((tif, run[0], run[1], out))
    if runs:
        import ctypes

        from . import _native

        n = len(runs)
        fds = np.fromiter((r[0].fileno() for r in runs), dtype=np.int32, count=n)
        offs = np.fromiter((r[1] for r in runs), dtype=np.int64, count=n)
        lens = np.fromiter((r[2] for r in runs), dtype=np.int64, count=n)
        dsts = np.fromiter((r[3].ctypes.data for r in runs), dtype=np.uint64, count=n)
        failed = (ctypes.c_int64 * 2)(-1, 0)
        rc = _native.lib().mg_host_read_runs(fds.ctypes.data, offs.ctypes.data, lens.ctypes.data, dsts.ctypes.data, n,
                                             max(1, min(int(workers), 64)), ctypes.addressof(failed))
        if rc == -3:
            tif, off, nbytes, _ = runs[int(failed[0])]
            why = os.strerror(int(failed[1])) if failed[1] else "short read"
            raise TiffError(f"{tif.path}: {why} in the {nbytes} bytes at {off}")
        _native.check(rc, "mg_host_read_runs")
    if slow:
        if workers > 1 and len(slow) > 1:
            from concurrent.futures import ThreadPoolExecutor

            with ThreadPoolExecutor(max_workers=int(workers)) as pool:
                for job in [pool.submit(tif.read_page_into, index, out) for tif, index, out in slow]:
                    job.result()  # (re-raises what a reader thread met)
        else:
            for tif, index, out in slow:
                tif.read_page_into(index, out)


def _unpredict(block, predictor):
    if predictor == 1:
        return block
    if predictor == 2:  # horizontal differencing, modulo the sample width
        return np.cumsum(block, axis=1, dtype=block.dtype.newbyteorder("="))
    raise TiffError(f"predictor {predictor} is not supported")


def _decode(data, compression, expected, path):
    if compression == 1:
        out = data
    elif compression in (8, 32946):
        try:
            out = zlib.decompress(data)
        except zlib.error as exc:
            raise TiffError(f"{path}: corrupt Deflate segment: {exc}") from None
    elif compression == 32773:
        out = _unpackbits(data)
    elif compression == 5:
        out = _unlzw(data)
    else:
        raise TiffError(f"{path}: TIFF compression {compression} is not supported")
    if len(out) < expected:
        raise TiffError(f"{path}: a segment decodes to {len(out)} bytes, {expected} expected")
    return out


def _unpackbits(data):
    out, i, n = bytearray(), 0, len(data)
    while i < n:
        c = data[i]
        i += 1
        if c < 128:
            out += data[i: i + c + 1]
            i += c + 1
        elif c > 128:
            out += data[i: i + 1] * (257 - c)
            i += 1
    return bytes(out)


def _unlzw(data):
    """TIFF's LZW (MSB-first codes, 9..12 bits, early change; Clear = 256, EOI = 257)."""
    table = [bytes([i]) for i in range(256)] + [b"", b""]
    out, bits, nbits, width, prev = bytearray(), 0, 0, 9, None
    for byte in data:
        bits = (bits << 8) | byte
        nbits += 8
        while nbits >= width:
            code = (bits >> (nbits - width)) & ((1 << width) - 1)
            nbits -= width
            bits &= (1 << nbits) - 1  # keep only what has not been consumed (else every shift is O(strip))
            if code == 256:
                del table[258:]
                width, prev = 9, None
                continue
            if code == 257:
                return bytes(out)
            if prev is None:
                entry = table[code]
            elif code < len(table):
                entry = table[code]
                table.append(prev + entry[:1])
            elif code == len(table):
                entry = prev + prev[:1]
                table.append(entry)
            else:
                raise TiffError("corrupt LZW segment")
            out += entry
            prev = entry
            if len(table) >= (1 << width) - 1 and width < 12:
                width += 1
    return bytes(out)
