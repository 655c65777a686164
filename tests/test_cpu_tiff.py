"""The TIFF layer (magnify_amd/tiff.py) and the reader's metadata rules (reference: src/magnify/reader.py:163-324) on
hand-packed files (tests/tiffwrite.py) and on files written by Pillow: classic and BigTIFF, both byte orders, strips and
tiles, codecs, OME-XML axes, MicroManager StartTime / ChNames, ImageJ hyperstacks, the reference's refusals."""
import datetime

import numpy as np
import pytest
from tiffwrite import ome_xml, write_tiff

from magnify_amd import reader, tiff


def _pages(n, shape=(37, 53), dtype=np.uint16, seed=0):
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "f":
        return [rng.random(shape).astype(dtype) for _ in range(n)]
    return [rng.integers(0, np.iinfo(dtype).max, shape, endpoint=True).astype(dtype) for _ in range(n)]


@pytest.mark.parametrize("bigtiff", [False, True])
@pytest.mark.parametrize("byteorder", ["<", ">"])
@pytest.mark.parametrize("layout", [{}, {"rows_per_strip": 5}, {"tile": (16, 16)}, {"compression": 8, "rows_per_strip": 7},
                                    {"compression": 32773}, {"compression": 8, "predictor": 2, "tile": (16, 32)},
                                    {"compression": 32946, "predictor": 2}])
def test_roundtrip_layouts(tmp_path, bigtiff, byteorder, layout):
    pages = _pages(5)
    path = tmp_path / "x.tif"
    write_tiff(path, pages, bigtiff=bigtiff, byteorder=byteorder, **layout)
    with tiff.TiffFile(path) as tif:
        assert tif.big == bigtiff and len(tif) == 5
        assert tif.axes == "IYX" and tif.shape == (5, 37, 53)
        for i in (3, 0, 4, 1, 2):  # any order: pages are addressed, not streamed
            np.testing.assert_array_equal(tif.asarray(i), pages[i])
        assert (tif.page(0).contiguous is not None) == (layout in ({}, {"rows_per_strip": 5}))
        with pytest.raises(IndexError):
            tif.page(5)


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.uint32, np.int16, np.float32, np.float64])
def test_sample_types(tmp_path, dtype):
    pages = _pages(2, dtype=dtype)
    write_tiff(tmp_path / "t.tif", pages, bigtiff=True, byteorder=">")
    with tiff.TiffFile(tmp_path / "t.tif") as tif:
        assert tif.dtype == np.dtype(dtype)
        np.testing.assert_array_equal(tif.asarray(1), pages[1])
        out = np.empty((37, 53), dtype=np.int8)
        with pytest.raises(tiff.TiffError):
            tif.read_page_into(0, out)


def test_pillow_written_files(tmp_path):
    """Files of an independent writer (libtiff through Pillow): LZW, Adobe Deflate, PackBits with replicate runs, raw."""
    from PIL import Image

    a = (np.arange(64 * 48) % 6553).astype(np.uint16).reshape(48, 64)
    a[10:20] = 7
    for codec, number in (("tiff_lzw", 5), ("tiff_adobe_deflate", 8), ("packbits", 32773), ("raw", 1)):
        Image.fromarray(a).save(tmp_path / "p.tif", compression=codec)
        with tiff.TiffFile(tmp_path / "p.tif") as tif:
            assert tif.page(0).compression == number
            np.testing.assert_array_equal(tif.asarray(0), a)
            assert tif.axes == "YX"
    # and Pillow reads a hand-packed classic file (the writer of these tests is a valid one)
    pages = _pages(3)
    write_tiff(tmp_path / "w.tif", pages)
    im = Image.open(tmp_path / "w.tif")
    assert im.n_frames == 3
    im.seek(2)
    np.testing.assert_array_equal(np.array(im), pages[2])


def test_malformed_files(tmp_path):
    (tmp_path / "a.tif").write_bytes(b"not a tiff at all")
    with pytest.raises(tiff.TiffError):
        tiff.TiffFile(tmp_path / "a.tif")
    write_tiff(tmp_path / "b.tif", _pages(1), bigtiff=True)
    raw = bytearray((tmp_path / "b.tif").read_bytes())
    (tmp_path / "c.tif").write_bytes(raw[: len(raw) // 2])  # IFD beyond the end
    with pytest.raises(tiff.TiffError):
        tiff.TiffFile(tmp_path / "c.tif").page(0)


@pytest.mark.parametrize("order,axes", [("XYCZT", "TCYX"), ("XYTZC", "CTYX"), ("XYZCT", "TCYX")])
def test_ome_bigtiff_axes(tmp_path, order, axes):
    """OME-XML in a BigTIFF: series axes = the reversed DimensionOrder without singletons (tifffile's squeezed axes)."""
    n_c, n_t = 2, 3
    pages = _pages(n_c * n_t)
    write_tiff(tmp_path / "o.ome.tif", pages, bigtiff=True,
               description=ome_xml(size_c=n_c, size_t=n_t, size_y=37, size_x=53, order=order, channel_names=["dapi", "cy5"]))
    with tiff.TiffFile(tmp_path / "o.ome.tif") as tif:
        assert tif.axes == axes
        assert tif.shape == tuple({"T": n_t, "C": n_c}[a] for a in axes[:2]) + (37, 53)
        assert tif.ome_channel_names == ["dapi", "cy5"]
    lay = reader.series_layout(tmp_path / "o.ome.tif")
    assert lay["dims"] == [{"T": "time", "C": "channel"}[a] for a in axes[:2]] and lay["page"] == (37, 53)
    # through the Reader: the in-file axes land in the standard order, page k of the file is C-order over the axes
    xp = list(reader.Reader()(str(tmp_path / "o.ome.tif")))[0]
    assert xp.tile.dims == ("channel", "time", "tile_y", "tile_x")
    for c in range(n_c):
        for t in range(n_t):
            k = t * n_c + c if axes == "TCYX" else c * n_t + t
            np.testing.assert_array_equal(xp.tile.values[c, t], pages[k])


def test_ome_refusals_and_positions(tmp_path):
    pages = _pages(4)
    # a Z axis: refused (reader.py:256-257)
    write_tiff(tmp_path / "z.ome.tif", pages, bigtiff=True, description=ome_xml(size_z=4, size_y=37, size_x=53))
    with pytest.raises(ValueError, match="Z dimension"):
        list(reader.Reader()(str(tmp_path / "z.ome.tif")))
    # positions (several Image elements in one file): the R axis is ignored, the first position is read (reader.py:249-254)
    write_tiff(tmp_path / "r.ome.tif", pages, bigtiff=True, description=ome_xml(size_t=2, size_y=37, size_x=53, n_images=2))
    with tiff.TiffFile(tmp_path / "r.ome.tif") as tif:
        assert tif.axes == "RTYX" and tif.shape == (2, 2, 37, 53)
    xp = list(reader.Reader()(str(tmp_path / "r.ome.tif")))[0]
    assert xp.tile.dims == ("time", "tile_y", "tile_x")
    np.testing.assert_array_equal(xp.tile.values, np.stack(pages[:2]))
    # a dimension named in the path AND inside the file (reader.py:260-262)
    write_tiff(tmp_path / "s_20240101.ome.tif", pages, bigtiff=True, description=ome_xml(size_t=4, size_y=37, size_x=53))
    with pytest.raises(ValueError, match="overlap"):
        list(reader.Reader()(str(tmp_path / "s_(time|%Y%m%d).ome.tif")))
    # undescribed pages: the reference has no dimension for tifffile's "I" axis
    write_tiff(tmp_path / "plain.tif", pages)
    with pytest.raises(ValueError, match="no description of their axes"):
        list(reader.Reader()(str(tmp_path / "plain.tif")))


def test_micromanager_times_and_channels(tmp_path):
    """reader.py:211-247: StartTime (time zone cut off) + Plane DeltaT in ms, every SizeC-th plane; ChNames."""
    n_c, n_t = 2, 3
    pages = _pages(n_c * n_t)
    deltas = [0.0, 4.0, 60000.0, 60004.5, 120000.0, 120004.0]
    summary = {"StartTime": "2024-03-05 10:20:30.500 -0800", "ChNames": ["bf", "egfp"]}
    write_tiff(tmp_path / "mm.ome.tif", pages, mm_summary=summary, mm_page_tag=True,
               description=ome_xml(size_c=n_c, size_t=n_t, size_y=37, size_x=53, delta_t_ms=deltas))
    with tiff.TiffFile(tmp_path / "mm.ome.tif") as tif:
        assert tif.is_micromanager and tif.micromanager_metadata["Summary"] == summary
    xp = list(reader.Reader()(str(tmp_path / "mm.ome.tif")))[0]
    start = datetime.datetime(2024, 3, 5, 10, 20, 30, 500000)
    want = [int((start + datetime.timedelta(milliseconds=d)).timestamp()) for d in deltas[::n_c]]
    assert list(xp.coords["time"].values) == want
    assert list(xp.coords["channel"].values) == ["bf", "egfp"]
    # channel named in the path: ChNames is not consulted, time still comes from the file
    write_tiff(tmp_path / "t_a.ome.tif", pages[:3], mm_summary=summary, mm_page_tag=True,
               description=ome_xml(size_t=3, size_y=37, size_x=53, delta_t_ms=[0, 1000, 2000]))
    write_tiff(tmp_path / "t_b.ome.tif", pages[3:], mm_summary=summary, mm_page_tag=True,
               description=ome_xml(size_t=3, size_y=37, size_x=53, delta_t_ms=[0, 1000, 2000]))
    xp = list(reader.Reader()(str(tmp_path / "t_(channel).ome.tif")))[0]
    assert xp.tile.dims == ("channel", "time", "tile_y", "tile_x") and list(xp.coords["channel"].values) == ["a", "b"]
    assert list(xp.coords["time"].values) == [int((start + datetime.timedelta(seconds=k)).timestamp()) for k in range(3)]
    np.testing.assert_array_equal(xp.tile.values[1, 2], pages[5])
    # without the MicroManager page tag the summary block is not looked at (tifffile's is_micromanager)
    write_tiff(tmp_path / "no.ome.tif", pages, description=ome_xml(size_c=n_c, size_t=n_t, size_y=37, size_x=53))
    xp = list(reader.Reader()(str(tmp_path / "no.ome.tif")))[0]
    assert "time" not in xp.coords and "channel" not in xp.coords


def test_stream_tiled_ome_bigtiff_series(tmp_path):
    """config C5's file shape at test scale: one OME-BigTIFF per tile position holding (time, channel) pages, streamed by
    iter_time_chunks chunk by chunk -- equal to the eager Reader and to the arrays that were written."""
    n_t, n_c, rows, cols, ty, tx = 5, 2, 2, 3, 24, 40
    rng = np.random.default_rng(3)
    data = rng.integers(0, 65535, (n_t, n_c, rows, cols, ty, tx)).astype(np.uint16)
    for r in range(rows):
        for c in range(cols):
            pages = [data[t, ch, r, c] for t in range(n_t) for ch in range(n_c)]
            write_tiff(tmp_path / f"acq_r{r}_c{c}.ome.tif", pages, bigtiff=True, rows_per_strip=8,
                       description=ome_xml(size_c=n_c, size_t=n_t, size_y=ty, size_x=tx, channel_names=["a", "b"]))
    pattern = str(tmp_path / "acq_r(row)_c(col).ome.tif")
    got = list(reader.iter_time_chunks(pattern, 2))
    assert [len(g[0]) for g in got] == [2, 2, 1] and got[0][1] == [0, 1]
    np.testing.assert_array_equal(np.concatenate([g[2] for g in got]), data)
    xp = list(reader.Reader()(pattern))[0]
    assert xp.tile.dims == ("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x")
    np.testing.assert_array_equal(xp.tile.values, data.transpose(1, 0, 2, 3, 4, 5))
