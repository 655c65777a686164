"""Hand-packed TIFF files for the tests and the C5 measurement: classic or BigTIFF, either byte order, strips or
tiles, optional Deflate / PackBits, an ImageDescription on the first page (OME-XML, ImageJ), MicroManager's summary
block and per-page metadata tag.  Written from the TIFF 6.0 / BigTIFF layout, independently of magnify_amd.tiff (which
only reads)."""
from __future__ import annotations

import json
import struct
import zlib

import numpy as np


def _packbits(data: bytes) -> bytes:
    out, i, n = bytearray(), 0, len(data)
    while i < n:  # literal runs only (valid PackBits; the reader's replicate branch is covered by Pillow-written files)
        chunk = data[i: i + 128]
        out.append(len(chunk) - 1)
        out += chunk
        i += len(chunk)
    return bytes(out)


def write_tiff(path, pages, bigtiff=False, byteorder="<", description=None, rows_per_strip=None, tile=None,
               compression=1, predictor=1, mm_summary=None, mm_page_tag=False, sample_format=None):
    """pages: sequence of equally typed 2-D arrays.  ``tile=(th, tw)`` writes tiles instead of strips."""
    bo = byteorder
    off_fmt, off_size = ("Q", 8) if bigtiff else ("I", 4)
    buf = bytearray()
    buf += (b"II" if bo == "<" else b"MM")
    if bigtiff:
        buf += struct.pack(bo + "HHHQ", 43, 8, 0, 0)
    else:
        buf += struct.pack(bo + "HI", 42, 0)
    if mm_summary is not None:
        if bigtiff:
            raise ValueError("MicroManager's header block belongs to classic TIFF")
        js = json.dumps(mm_summary).encode()
        buf += struct.pack(bo + "8I", 54773648, 0, 483765892, 0, 99384722, 0, 2355492, len(js)) + js
        if len(buf) % 2:
            buf += b"\x00"
    first_ifd_field = 8 if bigtiff else 4
    prev_next_field = first_ifd_field
    for n, page in enumerate(pages):
        page = np.ascontiguousarray(page)
        h, w = page.shape
        dt = page.dtype.newbyteorder(bo)
        raw = page.astype(dt, copy=False)

        def diff(block):  # horizontal differencing inside a strip / tile (predictor 2)
            if predictor != 2:
                return block
            d = block.astype(block.dtype.newbyteorder("="))
            d[:, 1:] = d[:, 1:] - d[:, :-1]
            return d.astype(dt)

        segments = []
        if tile:
            th, tw = tile
            for r0 in range(0, h, th):
                for c0 in range(0, w, tw):
                    t = np.zeros((th, tw), dtype=dt)
                    blk = raw[r0: r0 + th, c0: c0 + tw]
                    t[: blk.shape[0], : blk.shape[1]] = blk
                    segments.append(diff(t).tobytes())
        else:
            rps = rows_per_strip or h
            for r0 in range(0, h, rps):
                segments.append(diff(raw[r0: r0 + rps]).tobytes())
        if compression in (8, 32946):
            segments = [zlib.compress(s) for s in segments]
        elif compression == 32773:
            segments = [_packbits(s) for s in segments]
        elif compression != 1:
            raise ValueError("write_tiff: compression 1, 8, 32946 or 32773")
        offsets = []
        for s in segments:
            if len(buf) % 2:
                buf += b"\x00"
            offsets.append(len(buf))
            buf += s
        counts = [len(s) for s in segments]
        kind = {"u": 1, "i": 2, "f": 3}[page.dtype.kind] if sample_format is None else sample_format
        tags = [(256, 4, [w]), (257, 4, [h]), (258, 3, [page.dtype.itemsize * 8]), (259, 3, [compression]), (262, 3, [1]),
                (277, 3, [1]), (284, 3, [1]), (339, 3, [kind])]
        if predictor != 1:
            tags.append((317, 3, [predictor]))
        big_t = 16 if bigtiff else 4
        if tile:
            tags += [(322, 4, [tile[1]]), (323, 4, [tile[0]]), (324, big_t, offsets), (325, big_t, counts)]
        else:
            tags += [(273, big_t, offsets), (278, 4, [rows_per_strip or h]), (279, big_t, counts)]
        if n == 0 and description is not None:
            tags.append((270, 2, description.encode("utf-8") + b"\x00"))
        if mm_page_tag:
            tags.append((51123, 2, json.dumps({"FrameIndex": n}).encode() + b"\x00"))
        tags.sort(key=lambda t: t[0])
        # out-of-line values first, then the IFD
        entries = []
        for tag, typ, vals in tags:
            if typ == 2:
                data, count = bytes(vals), len(vals)
            else:
                fmt = {3: "H", 4: "I", 16: "Q"}[typ]
                data, count = struct.pack(bo + fmt * len(vals), *vals), len(vals)
            if len(data) <= off_size:
                field = data + b"\x00" * (off_size - len(data))
            else:
                if len(buf) % 2:
                    buf += b"\x00"
                field = struct.pack(bo + off_fmt, len(buf))
                buf += data
            entries.append((tag, typ, count, field))
        if len(buf) % 2:
            buf += b"\x00"
        ifd_at = len(buf)
        struct.pack_into(bo + off_fmt, buf, prev_next_field, ifd_at)
        if bigtiff:
            buf += struct.pack(bo + "Q", len(entries))
            for tag, typ, count, field in entries:
                buf += struct.pack(bo + "HHQ", tag, typ, count) + field
        else:
            buf += struct.pack(bo + "H", len(entries))
            for tag, typ, count, field in entries:
                buf += struct.pack(bo + "HHI", tag, typ, count) + field
        prev_next_field = len(buf)
        buf += struct.pack(bo + off_fmt, 0)
    with open(path, "wb") as fh:
        fh.write(buf)


def ome_xml(size_c=1, size_t=1, size_z=1, size_y=1, size_x=1, order="XYCZT", pixel_type="uint16", delta_t_ms=None,
            channel_names=None, n_images=1, file_name=None):
    """A minimal OME-XML block (2016-06 schema names).  ``delta_t_ms``: list of DeltaT per plane, in the order of the
    planes (DimensionOrder)."""
    images = []
    for i in range(n_images):
        chans = "".join(f'<Channel ID="Channel:{i}:{c}" Name="{(channel_names or [])[c] if channel_names else "ch%d" % c}" SamplesPerPixel="1"/>'
                        for c in range(size_c))
        planes = ""
        if delta_t_ms is not None:
            dims = order[2:]
            sizes = {"C": size_c, "T": size_t, "Z": size_z}
            for p, dt in enumerate(delta_t_ms):
                idx, rest = {}, p
                for a in dims:  # first listed varies fastest
                    idx[a] = rest % sizes[a]
                    rest //= sizes[a]
                planes += f'<Plane DeltaT="{dt}" DeltaTUnit="ms" TheC="{idx["C"]}" TheT="{idx["T"]}" TheZ="{idx["Z"]}"/>'
        uuid = f'<UUID FileName="{file_name}">urn:uuid:0</UUID>' if file_name else ""
        images.append(f'<Image ID="Image:{i}" Name="pos{i}"><Pixels ID="Pixels:{i}" DimensionOrder="{order}" Type="{pixel_type}" '
                      f'SizeC="{size_c}" SizeT="{size_t}" SizeZ="{size_z}" SizeY="{size_y}" SizeX="{size_x}">{chans}'
                      f'<TiffData IFD="{i * size_c * size_t * size_z}" PlaneCount="{size_c * size_t * size_z}">{uuid}</TiffData>{planes}</Pixels></Image>')
    return ('<?xml version="1.0" encoding="UTF-8"?><OME xmlns="http://www.openmicroscopy.org/Schemas/OME/2016-06">'
            + "".join(images) + "</OME>")
