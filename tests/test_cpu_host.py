"""CPU-only tests: the C-ABI library loads and exports every declared symbol, host tables match the
oracle, the np.quantile restatement, the registry / Pipeline semantics of the reference
(pipeline.py:31-87, registry.py:16-29), the container, and format round trips."""
import ctypes
import os
import re

import numpy as np
import pytest

import magnify_amd as mg
from magnify_amd import _native as nat
from magnify_amd import hotpath as hp
from oracle import ref_numeric as rn
from oracle import ref_opencv as rcv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "magnify_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t) (mg_\w+)\(", header, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(nat.PROTOTYPES), declared ^ set(nat.PROTOTYPES)
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert nat.lib().mg_version() >= 1
    # the ctypes prototypes take as many arguments as the header declares (one short sends a pointer as an int)
    plain = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for m in re.finditer(r"\b(?:int|int64_t)\s+(mg_\w+)\s*\(([^;]*?)\)\s*;", plain, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert len(nat.PROTOTYPES[name]) == n, (name, n, len(nat.PROTOTYPES[name]))


def test_product_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        hp.require_gpu()
    with pytest.raises(RuntimeError):
        mg.beads(mg.DataArray(np.zeros((64, 64), np.uint16), ("y", "x")), overlap=0, num_iter=10)


def test_host_tables_match_oracle():
    for r in (0, 1, 2, 5, 10, 25, 33, 60):
        np.testing.assert_array_equal(nat.circle_points(r), rn.circle_points(r))
        np.testing.assert_array_equal(nat.circle_points(r, True), rn.circle_points(r, True))
    for r in range(2, 70):
        hw = nat.disk_halfwidths(r)
        pts = {tuple(p) for p in rn.filled_circle_points(r).tolist()}
        assert pts == {(dy, dx) for dy in range(-r, r + 1) for dx in range(-hw[dy + r], hw[dy + r] + 1)}
        np.testing.assert_array_equal(np.sort(mg.utils.filled_circle_points(r), axis=0),
                                      np.sort(rn.filled_circle_points(r), axis=0))
    for r in range(0, 40):
        hw = nat.cv_disk_halfwidths(r)
        mask = rcv.filled_circle_mask((2 * r + 1, 2 * r + 1), (r, r), r)
        want = np.zeros_like(mask)
        for ady in range(r + 1):
            want[[r - ady, r + ady], r - hw[ady] : r + hw[ady] + 1] = True
        np.testing.assert_array_equal(mask, want)
    rc, expected, starts = nat.perimeter_table(5, 25)
    assert starts[0] == 0 and starts[-1] == len(rc) == len(expected)
    np.testing.assert_array_equal(rc[starts[5] : starts[6]], rn.circle_points(10))
    # libm atan2 (product) vs NumPy's SIMD arctan2 (oracle): within 1 ulp of float64
    ref = np.arctan2(rc[:, 0], rc[:, 1])
    assert np.max(np.abs(expected - ref)) <= 4.5e-16
    with pytest.raises(ValueError):
        nat.disk_halfwidths(1)


def test_bounding_box_golden(golden):
    for x, y, length, w, h, *exp in golden("bounding_box")["cases"]:
        assert mg.utils.bounding_box(int(x), int(y), int(length), int(w), int(h)) == tuple(int(v) for v in exp)


@pytest.mark.parametrize("n", [1, 2, 7, 1000, 300 * 420, 5_000_001, 20_000_003])
def test_quantile_restatement_matches_numpy(n):
    """hotpath.quantile_indexes / lerp_f32 reproduce np.quantile(float32 array, q) bit-for-bit,
    including the float32 virtual index above 2**24 elements."""
    rng = np.random.default_rng(n)
    m = rng.integers(0, 5000, size=n) ** 2 + rng.integers(0, 5000, size=n) ** 2
    g = np.sqrt(m.astype(np.float32))
    srt = np.sort(m)
    for q in (0.0, 0.1, 0.5, 0.9, 0.99, 0.9997, 1.0):
        a, b, gamma = hp.quantile_indexes(n, q)
        got = hp.lerp_f32(hp._grad_of(int(srt[a])), hp._grad_of(int(srt[b])), gamma)
        want = np.quantile(g, q)
        assert got.dtype == np.float32 and got == want, (n, q, got, want)


def test_canny_threshold_prep():
    for lo, hi in [(0.0, 0.0), (3.5, 10.2), (10.2, 3.5), (40000.0, 50000.0), (0.5, 181.0193)]:
        assert hp.canny_int_thresholds(lo, hi) == rcv.canny_thresholds(lo, hi)


# ---- registry / pipeline --------------------------------------------------------------------


def test_registered_names_and_signatures():
    names = set(mg.components.get_all())
    assert {"standardize_format", "flatfield_correct", "stitch", "find_beads", "identify_buttons", "drop",
            "restore_format", "rotate"} <= names
    assert "read" in mg.readers.get_all()
    import inspect

    assert list(inspect.signature(mg.components.get("flatfield_correct")).parameters) == ["flatfield", "darkfield"]
    sig = inspect.signature(mg.beads)
    assert sig.parameters["min_bead_diameter"].default == 10 and sig.parameters["num_iter"].default == 5000000
    assert inspect.signature(mg.beads_pipe).parameters["min_bead_diameter"].default == 5  # registry.py:572
    assert inspect.signature(mg.microfluidic_chip).parameters["row_dist"].default == 375 / 1.61
    with pytest.raises(ValueError):
        mg.microfluidic_chip_pipe(chip_type="nope")


def test_component_decorator_returns_function_and_registers_factory():
    @mg.component("unit_test_scale")
    def scale(xp, factor=2):
        return ("scaled", xp, factor)

    assert scale("a") == ("scaled", "a", 2)  # the original function comes back (registry.py:27)
    factory = mg.components.get("unit_test_scale")
    assert factory(factor=3)("b") == ("scaled", "b", 3)


def test_pipeline_add_remove_semantics():
    pipe = mg.Pipeline("read")
    pipe.add_pipe("standardize_format")
    pipe.add_pipe("drop", roi_only=False)

    def custom(xp, k=1):
        return xp

    pipe.add_pipe(custom, after="standardize_format", k=2)
    pipe.add_pipe("rotate", first=True)
    pipe.add_pipe("restore_format", before="drop")
    assert [n for n, _ in pipe.components] == ["rotate", "standardize_format", "custom", "restore_format", "drop"]
    with pytest.raises(ValueError):
        pipe.add_pipe("rotate")  # duplicate name
    with pytest.raises(ValueError):
        pipe.add_pipe("rotate", name="r2", first=True, last=True)
    pipe.add_pipe("rotate", name="r2", after=0)
    assert pipe.components[1][0] == "r2"
    pipe.remove_pipe("r2")
    with pytest.raises(ValueError):
        pipe.remove_pipe("r2")
    with pytest.raises(ValueError):
        mg.Pipeline("read").remove_pipe("x")
    with pytest.raises(KeyError):
        pipe.add_pipe("no_such_component")


def test_reader_passthrough_and_errors():
    pipe = mg.Pipeline("read")
    a = mg.DataArray(np.zeros((4, 4)), ("y", "x"))
    assert pipe(a) is a
    out = pipe([a, a])
    assert isinstance(out, list) and len(out) == 2
    with pytest.raises(FileNotFoundError):
        pipe("/no/such/file.tif")


@pytest.mark.parametrize("dims", [("y", "x"), ("channel", "y", "x"), ("time", "y", "x"), ("channel", "time", "y", "x"),
                                  ("time", "channel", "y", "x"), ("row", "col", "y", "x")])
def test_standardize_restore_round_trip(dims):
    rng = np.random.default_rng(0)
    shape = tuple(2 + i for i in range(len(dims) - 2)) + (6, 7)
    data = rng.integers(0, 100, size=shape).astype(np.uint16)
    xp = mg.preprocess.standardize_format(mg.DataArray(data, dims))
    assert xp.tile.dims == ("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x")
    assert xp.attrs["__original_tile_dims__"] == ["tile_" + d if d in ("x", "y", "row", "col") else d for d in dims]
    back = mg.postprocess.restore_format(xp)
    assert "__original_tile_dims__" not in back.attrs
    np.testing.assert_array_equal(back.tile.values, data)
    assert back.tile.dims == tuple("tile_" + d if d in ("x", "y", "row", "col") else d for d in dims)


def test_standardize_stacks_extra_dims_into_time():
    data = np.arange(2 * 3 * 4 * 5).reshape(2, 3, 4, 5).astype(np.uint16)
    xp = mg.preprocess.standardize_format(mg.DataArray(data, ("well", "time", "y", "x")))
    assert xp.tile.sizes["time"] == 6 and xp.tile.sizes["channel"] == 1
    back = mg.postprocess.restore_format(xp)
    np.testing.assert_array_equal(back.tile.values, data)
    assert back.tile.dims == ("well", "time", "tile_y", "tile_x")


def test_identify_buttons_and_drop():
    xp = mg.preprocess.standardize_format(mg.DataArray(np.zeros((3, 8, 8), np.uint16), ("time", "y", "x")))
    xp = mg.identify.identify_buttons(xp, shape=(4, 5))
    assert xp.tag.shape == (4, 5) and (xp.tag.values == "default").all()
    assert xp.valid.shape == (4, 5, 3) and xp.valid.values.all()
    with pytest.raises(ValueError):
        mg.identify.identify_buttons(xp)
    assert "tile" not in mg.postprocess.drop(xp).data_vars
    assert "tile" in mg.postprocess.drop(xp, drop_tiles=False).data_vars


def test_container_algebra():
    rng = np.random.default_rng(1)
    roi = mg.DataArray(rng.integers(0, 50, size=(3, 2, 4, 4)).astype(np.uint16), ("mark", "channel", "roi_y", "roi_x"))
    fg = mg.DataArray(rng.random((3, 4, 4)) > 0.5, ("mark", "roi_y", "roi_x"))
    ds = mg.Dataset({"roi": roi}, coords={"fg": fg, "channel": ["a", "b"]})
    want = np.where(fg.values[:, None], roi.values, np.nan)
    np.testing.assert_allclose(ds.roi.where(ds.fg).mean(dim=["roi_x", "roi_y"]).values, np.nanmean(want, axis=(-1, -2)))
    np.testing.assert_allclose(ds.where(ds.fg).roi.median(dim=["roi_x", "roi_y"]).values,
                               np.nanmedian(want, axis=(-1, -2)))
    np.testing.assert_array_equal(ds.fg.sum(dim=["roi_x", "roi_y"]).values, fg.values.sum(axis=(-1, -2)))
    np.testing.assert_array_equal(ds.roi.sel(channel="b").values, roi.values[:, 1])
    np.testing.assert_array_equal(ds.roi.isel(mark=[0, 2]).values, roi.values[[0, 2]])
    np.testing.assert_array_equal(ds.roi[1, 0, 1:3].values, roi.values[1, 0, 1:3])
    assert ds.sizes == {"mark": 3, "channel": 2, "roi_y": 4, "roi_x": 4}


def test_mark_stack_unstack():
    x = mg.DataArray(np.arange(24).reshape(2, 3, 4), ("mark_row", "mark_col", "time"))
    tag = mg.DataArray(np.array([["a", "b", "c"], ["d", "e", "f"]]), ("mark_row", "mark_col"))
    ds = mg.Dataset(coords={"x": x, "tag": tag})
    st = ds.stack_mark()
    assert st.x.dims == ("mark", "time") and st.sizes["mark"] == 6
    assert st.tag.values.tolist() == ["a", "b", "c", "d", "e", "f"]
    back = st.unstack()
    np.testing.assert_array_equal(back.x.values, x.values)
    assert back.x.dims == ("mark_row", "mark_col", "time")


def test_save_load_roundtrip(tmp_path):
    """mg.save / mg.load (file.py:6-17): NetCDF-3 with xarray's encoding conventions; a chip dataset
    is unstacked on save and restacked on load."""
    rng = np.random.default_rng(0)
    nr, nc, L = 3, 4, 6
    roi = rng.integers(0, 65535, size=(nr, nc, 2, 1, L, L)).astype(np.uint16)
    fg = rng.random((nr, nc, 1, L, L)) > 0.5
    tag = np.array([["a", "", "gfp-long-name", "x"]] * nr)
    ds = mg.Dataset({"roi": mg.DataArray(roi, ("mark_row", "mark_col", "channel", "time", "roi_y", "roi_x")),
                     "image": mg.DataArray(rng.integers(0, 9, (2, 1, 8, 9)).astype(np.uint16), ("channel", "time", "im_y", "im_x"))},
                    coords={"fg": (("mark_row", "mark_col", "time", "roi_y", "roi_x"), fg),
                            "x": (("mark_row", "mark_col", "time"), rng.random((nr, nc, 1))),
                            "tag": (("mark_row", "mark_col"), tag), "channel": ["egfp", "dna"],
                            "time": np.array([7], dtype=np.int64)},
                    attrs={"name": "chip 1", "overlap": 102})
    stacked = ds.stack_mark()
    path = tmp_path / "assay.nc"
    mg.save(path, stacked)
    back = mg.load(path)
    assert back.roi.dims[0] == "mark" and back.roi.shape == (nr * nc, 2, 1, L, L)
    np.testing.assert_array_equal(back.roi.values, stacked.roi.values)
    assert back.roi.dtype == np.uint16 and back.fg.dtype == np.bool_
    np.testing.assert_array_equal(back.fg.values, stacked.fg.values)
    np.testing.assert_array_equal(back.x.values, stacked.x.values)
    np.testing.assert_array_equal(back.tag.values, stacked.tag.values)
    np.testing.assert_array_equal(back.image.values, ds.image.values)
    assert list(back.channel.values) == ["egfp", "dna"] and int(back.time.values[0]) == 7
    assert back.attrs["name"] == "chip 1" and int(back.attrs["overlap"]) == 102
    assert "fg" in back.coords and "roi" in back.data_vars and "tag" in back.coords
    # the file is a plain NetCDF-3 that follows xarray's conventions
    from scipy.io import netcdf_file

    with netcdf_file(str(path), "r", mmap=False) as f:
        assert f.variables["roi"].dimensions[:2] == ("mark_row", "mark_col")
        assert f.variables["roi"]._attributes["_Unsigned"] in ("true", b"true")
        assert f.variables["fg"]._attributes["dtype"] in ("bool", b"bool")
        assert f.variables["tag"].dimensions[-1].startswith("string")
    with pytest.raises(ValueError):
        mg.save(tmp_path / "big.nc", mg.Dataset({"v": mg.DataArray(np.array([2**40]), ("n",))}))


def _write_tiff(path, arr, pages=None, description=None):
    from PIL import Image

    if pages is None:
        Image.fromarray(arr).save(path)
    else:
        ims = [Image.fromarray(p) for p in pages]
        ims[0].save(path, save_all=True, append_images=ims[1:], description=description)


def test_reader_path_patterns(tmp_path):
    """The reference's pattern grammar (reader.py:80-160) and tile assembly (reader.py:163-324) on
    TIFF files written with Pillow."""
    from magnify_amd import reader

    rng = np.random.default_rng(1)
    truth = {}
    for assay in ("chipA", "chipB"):
        for ch in ("egfp", "dna"):
            for day, conc in (("20240102", "0.5"), ("20240105", "2.0")):
                for r in range(2):
                    for c in range(3):
                        img = rng.integers(0, 60000, (5, 7)).astype(np.uint16)
                        truth[assay, ch, day, r, c] = img
                        _write_tiff(tmp_path / f"{assay}_{ch}_{day}_{conc}uM_r{r}c{c}.tif", img)
    pattern = str(tmp_path / "(assay)_(channel)_(time|%Y%m%d)_(conc_time|float)uM_r(row)c(col).tif")
    assays = list(reader.Reader()(pattern))
    assert [a.attrs["name"] for a in assays] == ["chipA", "chipB"]
    xp = assays[1]
    assert xp.tile.dims == ("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x")
    assert xp.tile.shape == (2, 2, 2, 3, 5, 7) and xp.tile.dtype == np.uint16
    assert list(xp.channel.values) == ["dna", "egfp"]
    import datetime

    assert list(xp.time.values) == [int(datetime.datetime(2024, 1, 2).timestamp()), int(datetime.datetime(2024, 1, 5).timestamp())]
    assert list(xp.conc.values) == [0.5, 2.0] and xp.conc.dims == ("time",)
    for (assay, ch, day, r, c), img in truth.items():
        if assay == "chipB":
            np.testing.assert_array_equal(
                xp.tile.values[["dna", "egfp"].index(ch), ["20240102", "20240105"].index(day), r, c], img)
    # no assay / time groups: a nameless experiment with the dimensions that are in the pattern
    one = list(reader.Reader()(str(tmp_path / "chipA_(channel)_20240102_0.5uM_r(row)c(col).tif")))
    assert len(one) == 1 and one[0].attrs["name"] == "" and one[0].tile.dims == ("channel", "tile_row", "tile_col", "tile_y", "tile_x")
    with pytest.raises(FileNotFoundError):
        list(reader.Reader()(str(tmp_path / "nothing_(channel).tif")))
    with pytest.raises(ValueError):  # two files for one index
        list(reader.Reader()(str(tmp_path / "chipA_(channel)_*_r0c0.tif")))
    # ImageJ hyperstack: time and channel inside the file
    pages = [rng.integers(0, 255, (4, 6)).astype(np.uint8) for _ in range(6)]
    _write_tiff(tmp_path / "stack_r0c0.tif", None, pages, "ImageJ=1.53t\nimages=6\nchannels=2\nframes=3\nhyperstack=true")
    hs = list(reader.Reader()(str(tmp_path / "stack_r(row)c(col).tif")))[0]
    assert hs.tile.dims == ("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x") and hs.tile.shape == (2, 3, 1, 1, 4, 6)
    np.testing.assert_array_equal(hs.tile.values[1, 2, 0, 0], pages[2 * 2 + 1])  # channel is the fastest page axis
    # and the pipeline entry point accepts the pattern (the stitch itself needs the GPU)
    import torch

    call = lambda: mg.image(str(tmp_path / "chipA_(channel)_20240102_0.5uM_r(row)c(col).tif"), overlap=0)  # noqa: E731
    if torch.cuda.is_available():
        assert call().image.shape == (2, 10, 21)
    else:
        with pytest.raises(RuntimeError):
            call()


def test_circle_mask_component_matches_cv_circle():
    """preprocess.py:136-153: the kept region is cv.circle's filled disk (oracle restatement), also
    when it runs over the image border; ``mask_inner`` inverts it."""
    import magnify_amd as mg
    from oracle import ref_opencv as ro

    rng = np.random.default_rng(3)
    img = rng.integers(1, 60000, (2, 90, 120), dtype=np.uint16)
    for center, diameter in (((40, 60), 50), ((5, 110), 41), ((89, 0), 30)):
        want = ro.filled_circle_mask((90, 120), center, diameter // 2)
        for inner in (False, True):
            xp = mg.Dataset({"image": mg.DataArray(img.copy(), ("channel", "im_y", "im_x"))})
            got = mg.components.get("circle_mask")(center=center, diameter=diameter, mask_inner=inner)(xp).image.values
            keep = ~want if inner else want
            np.testing.assert_array_equal(got, img * keep)
            assert got.dtype == np.uint16
    with pytest.raises(NotImplementedError):
        mg.components.get("basic_correct")()(None)


def test_perimeter_table_structure_for_orientation_windows():
    """mg_score_circles' prefilter hard-codes the orientation quarter of each symmetric point of a
    perimeter group (3, 2, 0, 1, 0, 1, 3, 2 for (x,y), (y,x), (-x,y), (-y,x), (x,-y), (y,-x), (-x,-y),
    (-y,-x)): that needs every group's first entry (dr, dc) = (x, y) to satisfy x > 0 > y, x < -y."""
    from magnify_amd import _native as nat

    rc, expected, starts = nat.perimeter_table(2, 64)
    quarter = [3, 2, 0, 1, 0, 1, 3, 2]
    for i in range(len(starts) - 1):
        pts = rc[starts[i]:starts[i + 1]]
        r = 2 + i
        assert [tuple(p) for p in pts[:4]] == [(0, -r), (-r, 0), (0, r), (r, 0)]
        body = pts[4:]
        n_groups, rest = divmod(len(body), 8)
        assert rest in (0, 4)
        for g in range(n_groups):
            grp = body[8 * g: 8 * g + 8]
            x, y = int(grp[0][0]), int(grp[0][1])
            assert x > 0 > y and x < -y
            derived = [(x, y), (y, x), (-x, y), (-y, x), (x, -y), (y, -x), (-x, -y), (-y, -x)]
            assert sorted(derived) == sorted(tuple(int(v) for v in p) for p in grp)
            for (dr, dc), k in zip(derived, quarter):
                theta = np.arctan2(dr, dc) % np.pi
                assert k * np.pi / 4 <= theta <= (k + 1) * np.pi / 4
        if rest:
            x, y = int(body[-4][0]), int(body[-4][1])
            assert abs(x) == abs(y)
            assert sorted([(x, y), (-x, -y), (-x, y), (x, -y)]) == sorted(tuple(int(v) for v in p) for p in body[-4:])


def test_reader_streams_time_chunks(tmp_path):
    """reader.iter_time_chunks (config C5 / SURVEY 8f N2): a (channel, time) series comes back chunk by
    chunk in time order, without the whole stack ever being held."""
    from magnify_amd import reader

    rng = np.random.default_rng(5)
    days = ["20240101", "20240103", "20240102", "20240110", "20240107"]
    truth = {}
    for day in days:
        for ch in ("b", "a"):
            truth[ch, day] = rng.integers(0, 60000, (6, 9)).astype(np.uint16)
            _write_tiff(tmp_path / f"s_{ch}_{day}.tif", truth[ch, day])
    pattern = str(tmp_path / "s_(channel)_(time|%Y%m%d).tif")
    chunks = list(reader.iter_time_chunks(pattern, 2))
    assert [c[2].shape for c in chunks] == [(2, 2, 6, 9), (2, 2, 6, 9), (1, 2, 6, 9)]
    assert all(c[1] == ["a", "b"] and c[2].dtype == np.uint16 for c in chunks)
    order = sorted(days)
    stamps = [t for c in chunks for t in c[0]]
    assert stamps == sorted(stamps) and len(stamps) == 5
    for k, day in enumerate(order):
        block = chunks[k // 2][2][k % 2]
        np.testing.assert_array_equal(block[0], truth["a", day])
        np.testing.assert_array_equal(block[1], truth["b", day])
    with pytest.raises(FileNotFoundError):
        next(reader.iter_time_chunks(str(tmp_path / "none_(channel)_(time|%Y%m%d).tif"), 2))
    with pytest.raises(ValueError):  # no time group
        next(reader.iter_time_chunks(str(tmp_path / "s_(channel)_20240101.tif"), 2))


def test_mrbles_code_assignment_matches_loop_restatement():
    """identify.py:88-234 (code assignment of identify_mrbles): the vectorised lattice fit + EM of
    magnify_amd.identify.assign_codes against the oracle's statement-by-statement loops, on synthetic
    lanthanide ratios -- 3 x 3 codes in two ratio dimensions, unequal code populations, a few outliers."""
    from magnify_amd.identify import _fit_levels, assign_codes
    from oracle import ref_identify as ri

    rng = np.random.default_rng(11)
    levels = [np.array([0.1, 0.35, 0.8]), np.array([0.05, 0.3, 0.55])]
    codes = np.array([[a, b] for a in levels[0] for b in levels[1]])
    pops = rng.integers(12, 40, len(codes))
    truth = np.repeat(np.arange(len(codes)), pops)
    scale, shift = np.array([1.7, 0.9]), np.array([0.04, 0.02])
    X = codes[truth] * scale + shift + rng.normal(0, 0.012, (len(truth), 2))
    X = np.concatenate([X, rng.uniform(0, 1.6, (6, 2))])  # strays
    ratios = np.column_stack([np.ones(len(X)), X])
    # 1-D fit alone
    for d in range(2):
        lv, cnt = np.unique(codes[:, d], return_counts=True)
        pts = np.sort(X[:, d])
        a1, p1 = _fit_levels(pts, lv, cnt, 40)
        a2, p2 = ri.fit_1d(pts, lv, cnt, 40)
        assert (a1, p1) == pytest.approx((a2, p2), rel=0, abs=1e-12)
    tags, A, p = assign_codes(ratios, codes, n_grid=40)
    tags_o, A_o, p_o = ri.assign_codes(ratios, codes, n_grid=40)
    np.testing.assert_allclose(A, A_o, rtol=0, atol=1e-12)
    np.testing.assert_allclose(p, p_o, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(tags, tags_o)
    # and it decodes: nearly every clustered bead gets its code, most strays the outlier component
    assert (tags[: len(truth)] == truth).mean() > 0.97


def test_identify_buttons_pinlist(tmp_path):
    """identify.py:19-29: pin list rows "(col, row)" (1-based) -> tag[row, col]; blank names become ""."""
    csv = tmp_path / "pins.csv"
    csv.write_text('Indices,MutantID\n"(1,1)",wt\n"(2,1)",BLANK\n"(3,1)",m7\n"(1,2)",m2\n"(2,2)",blank\n"(3,2)",m9\n')
    xp = mg.preprocess.standardize_format(mg.DataArray(np.zeros((2, 8, 8), np.uint16), ("time", "y", "x")))
    xp = mg.identify.identify_buttons(xp, pinlist=str(csv))
    assert xp.tag.shape == (2, 3)
    assert xp.tag.values.tolist() == [["wt", "", "m7"], ["m2", "", "m9"]]
    assert xp.valid.shape == (2, 3, 2) and xp.valid.values.all()
    xp2 = mg.identify.identify_buttons(xp, pinlist=str(csv), blank=["wt"])
    assert xp2.tag.values[0, 0] == "" and xp2.tag.values[0, 1] == "BLANK"


def test_filter_nonround_border_lengths_and_component():
    """filter.py:40-62: outer-border length of simple shapes (values cv.arcLength of cv.findContours' external
    contours has for them: a filled s x s square 4 (s - 1), a one-pixel line there and back, a diagonal in
    sqrt(2) steps) and the component on a small Dataset: round disks stay, a sliver and an empty mask go."""
    from magnify_amd.filter import outer_border_length

    assert outer_border_length(np.ones((5, 5), bool)) == 16.0
    line = np.zeros((3, 7), bool)
    line[1, 1:6] = True
    assert outer_border_length(line) == 8.0
    assert outer_border_length(np.eye(6, dtype=bool)) == pytest.approx(10 * np.sqrt(2))
    one = np.zeros((3, 3), bool)
    one[1, 1] = True
    assert outer_border_length(one) == 0.0
    yy, xx = np.mgrid[-15:16, -15:16]
    disk = (yy * yy + xx * xx) <= 100
    sliver = np.zeros_like(disk)
    sliver[15, 4:27] = True
    two = disk.copy()
    two[:, 15] = False  # split into two half disks: two components, both counted
    fg = np.stack([disk, sliver, np.zeros_like(disk), two])[:, None]  # (mark, time, y, x)
    xp = mg.Dataset({"roi": mg.DataArray(np.zeros((4, 1, 1, 31, 31), np.uint16), ("mark", "channel", "time", "roi_y", "roi_x"))},
                    coords={"fg": (("mark", "time", "roi_y", "roi_x"), fg), "valid": (("mark", "time"), np.ones((4, 1), bool))})
    out = mg.components.get("filter_nonround")()(xp)
    assert out.valid.values[:, 0].tolist() == [True, False, False, False]
    roundness = 4 * np.pi * disk.sum() / outer_border_length(disk) ** 2
    assert 0.85 < roundness < 1.0
    assert mg.components.get("filter_nonround")(min_roundness=0.95)(xp).valid.values[0, 0] == False  # noqa: E712


def test_score_pair_table_bounds_every_term():
    """mg_score_pair_table (host side of the keyed scoring prefilter): the perimeter of every radius as pairs of
    opposite points -- together exactly the reference's circle_points set -- and, for every pair and every
    orientation bin, a signed byte that is an upper bound (1/64) of mean_grad's term 4 |d - pi/2| / pi - 1
    (utils.py:244-249) for ANY gradient angle whose orientation lies in the bin, at either point of the pair."""
    from magnify_amd import _native as nat

    table = nat.score_pair_table()
    assert table.shape == (27, 80)
    rng = np.random.default_rng(0)
    for r in range(2, 27):
        firsts = nat.score_pairs(r)
        n = len(firsts)
        rc, expected, _ = nat.perimeter_table(r, r)
        assert 2 * n == len(rc) <= 160
        both = np.concatenate([firsts, -firsts])
        assert {tuple(p) for p in both} == {tuple(p) for p in rc} and len({tuple(p) for p in both}) == 2 * n
        q = table[r, :n].view(np.int8).reshape(n, 8).astype(np.int64)
        assert (table[r, n:] == 0).all() and q.min() >= -48 and q.max() <= 64
        for sign in (1, -1):
            exp = np.arctan2(sign * firsts[:, 0].astype(np.float64), sign * firsts[:, 1].astype(np.float64))
            for b in range(8):
                # angles (either sign of the gradient) whose orientation falls into bin b, boundaries included
                phi = np.concatenate([np.linspace(b, b + 1, 41) * np.pi / 8, rng.uniform(b, b + 1, 200) * np.pi / 8])
                for ang in (phi, phi - np.pi):
                    d = np.abs(ang[None, :].astype(np.float32).astype(np.float64) - exp[:, None])
                    d = np.where(d > np.pi, d - np.pi, d)
                    term = 4 * np.abs(d - np.pi / 2) / np.pi - 1
                    ub = q[:, b, None] / 64.0
                    # (+2e-7: float32 angles at +-pi give terms of 1 + 1.1e-7 in the reference's own formula; the
                    # prefilter's threshold carries a 1e-3 margin for exactly this kind of rounding)
                    assert (term <= ub + 2e-7).all()
                    assert (ub - term.max(axis=1, keepdims=True) <= 1 / 64 + 1e-5).all()  # and not uselessly loose


def _bead_result_dataset():
    """A hand-built result with the reference's bead schema (find.py:503-555)."""
    import magnify_amd as mg

    m, c, t, L = 3, 2, 1, 8
    rng = np.random.default_rng(1)
    ds = mg.Dataset(attrs={"name": "assay0"})
    ds["image"] = mg.DataArray(rng.integers(0, 4000, (c, t, 32, 40)).astype(np.uint16), ("channel", "time", "im_y", "im_x"))
    ds["roi"] = mg.DataArray(rng.integers(0, 4000, (m, c, t, L, L)).astype(np.uint16), ("mark", "channel", "time", "roi_y", "roi_x"))
    ds.coords["channel"] = mg.DataArray(np.array(["egfp", "cy5"]), ("channel",), name="channel")
    ds.coords["fg"] = mg.DataArray(rng.random((m, t, L, L)) > 0.5, ("mark", "time", "roi_y", "roi_x"), name="fg")
    ds.coords["bg"] = mg.DataArray(rng.random((m, t, L, L)) > 0.5, ("mark", "time", "roi_y", "roi_x"), name="bg")
    ds.coords["x"] = mg.DataArray(rng.random((m, t)) * 40, ("mark", "time"), name="x")
    ds.coords["y"] = mg.DataArray(rng.random((m, t)) * 32, ("mark", "time"), name="y")
    ds.coords["valid"] = mg.DataArray(np.ones((m, t), dtype=bool), ("mark", "time"), name="valid")
    return ds


def test_to_xarray_with_real_xarray():
    """The xarray output surface against a REAL xarray (skipped where it is not installed -- neither this
    container nor the GPU box has it): Dataset.to_xarray() yields the reference's schema and a real
    xr.DataArray / xr.Dataset input is accepted by the components' entry adapter."""
    xr = pytest.importorskip("xarray")
    from magnify_amd import xr_lite

    ds = _bead_result_dataset()
    out = ds.to_xarray()
    assert isinstance(out, xr.Dataset)
    assert out["roi"].dims == ("mark", "channel", "time", "roi_y", "roi_x") and out["roi"].dtype == np.uint16
    assert out["image"].dims == ("channel", "time", "im_y", "im_x")
    for name, dtype in (("fg", bool), ("bg", bool), ("valid", bool), ("x", np.float64), ("y", np.float64)):
        assert name in out.coords and out.coords[name].dtype == dtype
    assert out.coords["fg"].dims == ("mark", "time", "roi_y", "roi_x") and out.coords["x"].dims == ("mark", "time")
    assert out.attrs["name"] == "assay0" and list(out.coords["channel"].values) == ["egfp", "cy5"]
    np.testing.assert_array_equal(out["roi"].values, ds["roi"].values)
    # user-side algebra of the README on the real object (README.md:21-22)
    np.testing.assert_allclose(out.roi.where(out.fg).mean(dim=["roi_x", "roi_y"]).values,
                               ds.roi.where(ds.fg).mean(dim=["roi_x", "roi_y"]).values)
    # and back in: real xarray inputs (all reference tests pass xr.DataArray, tests/test_beads.py:54)
    back = xr_lite.from_any(out)
    assert set(back.data_vars) == {"image", "roi"} and back["roi"].dims == ds["roi"].dims
    arr = xr_lite.from_any(xr.DataArray(np.zeros((4, 5), np.uint16), dims=("y", "x")))
    assert arr.dims == ("y", "x") and arr.shape == (4, 5)
    # chip results carry the (mark_row, mark_col) MultiIndex on mark (find.py:182-201, tests/test_chip.py:334-336)
    chip = _bead_result_dataset()
    chip.coords["mark_row"] = mg_dataarray(np.array([0, 0, 1]), ("mark",), "mark_row")
    chip.coords["mark_col"] = mg_dataarray(np.array([0, 1, 0]), ("mark",), "mark_col")
    chip._cache["mark_shape"] = (2, 2)
    xc = chip.to_xarray()
    assert "mark" in xc.indexes and list(xc.indexes["mark"].names) == ["mark_row", "mark_col"]


def mg_dataarray(values, dims, name):
    import magnify_amd as mg

    return mg.DataArray(values, dims, name=name)


def test_to_xarray_calls_with_the_reference_schema(monkeypatch):
    """Where xarray is absent: to_xarray() is executed against a recording stand-in for the xarray module, which
    checks WHAT is handed to xr.Dataset / xr.DataArray (names, dims, dtypes) -- the schema of SURVEY 8b."""
    import sys
    import types

    calls = {}

    class FakeDataset:
        def __init__(self, data_vars, coords=None, attrs=None):
            calls["dataset"] = (data_vars, coords, attrs)
            self.indexed = None

        def set_index(self, **kw):
            self.indexed = kw
            return self

    class FakeDataArray:
        def __init__(self, values, dims=None, coords=None, name=None, attrs=None):
            calls["dataarray"] = (values, dims, coords, name, attrs)

    fake = types.ModuleType("xarray")
    fake.Dataset, fake.DataArray = FakeDataset, FakeDataArray
    monkeypatch.setitem(sys.modules, "xarray", fake)
    ds = _bead_result_dataset()
    ds.attrs["__mg_private__"] = 1
    out = ds.to_xarray()
    data_vars, coords, attrs = calls["dataset"]
    assert set(data_vars) == {"image", "roi"} and attrs == {"name": "assay0"} and out.indexed is None
    assert data_vars["roi"][0] == ("mark", "channel", "time", "roi_y", "roi_x") and data_vars["roi"][1].dtype == np.uint16
    assert data_vars["image"][0] == ("channel", "time", "im_y", "im_x")
    assert coords["fg"][0] == ("mark", "time", "roi_y", "roi_x") and coords["fg"][1].dtype == bool
    assert coords["x"][0] == ("mark", "time") and coords["x"][1].dtype == np.float64
    assert coords["valid"][1].dtype == bool and list(coords["channel"][1]) == ["egfp", "cy5"]
    ds.roi.to_xarray()
    values, dims, acoords, name, _ = calls["dataarray"]
    assert dims == ("mark", "channel", "time", "roi_y", "roi_x") and values.dtype == np.uint16 and name == "roi"
    assert set(acoords) >= {"fg", "bg", "x", "y", "valid"}
    ds.coords["mark_row"] = mg_dataarray(np.array([0, 0, 1]), ("mark",), "mark_row")
    ds.coords["mark_col"] = mg_dataarray(np.array([0, 1, 0]), ("mark",), "mark_col")
    ds._cache["mark_shape"] = (2, 2)
    assert ds.to_xarray().indexed == {"mark": ("mark_row", "mark_col")}


def test_iter_time_chunks_tiles_and_hyperstacks(tmp_path):
    """reader.iter_time_chunks (the streaming half of SURVEY 8f N2): tiled series chunked by time, one page
    decoded at a time, equal to the eager Reader; ImageJ hyperstacks with time / channel inside the file."""
    from PIL import Image

    from magnify_amd import reader

    rng = np.random.default_rng(0)
    T, C, R, Cc, ty, tx = 3, 2, 2, 3, 16, 20
    data = rng.integers(0, 60000, (T, C, R, Cc, ty, tx)).astype(np.uint16)
    for t in range(T):
        for c in range(C):
            for r in range(R):
                for k in range(Cc):
                    Image.fromarray(data[t, c, r, k]).save(tmp_path / f"x_ch{c}_202401{t + 10}_r{r}_c{k}.tif")
    pattern = str(tmp_path / "x_(channel)_(time|%Y%m%d)_r(row)_c(col).tif")
    blocks = list(reader.iter_time_chunks(pattern, 2))
    assert [b[2].shape for b in blocks] == [(2, C, R, Cc, ty, tx), (1, C, R, Cc, ty, tx)] and blocks[0][1] == ["ch0", "ch1"]
    np.testing.assert_array_equal(np.concatenate([b[2] for b in blocks]), data)
    eager = list(reader.Reader()(pattern))[0]
    assert eager.tile.dims == ("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x")
    np.testing.assert_array_equal(np.asarray(eager.tile.values).transpose(1, 0, 2, 3, 4, 5), data)
    hs = rng.integers(0, 60000, (4, 2, 16, 20)).astype(np.uint16)
    ims = [Image.fromarray(hs[t, c]) for t in range(4) for c in range(2)]
    ims[0].save(tmp_path / "stack.tif", save_all=True, append_images=ims[1:],
                description="ImageJ=1.53\nimages=8\nchannels=2\nframes=4\nhyperstack=true")
    got = list(reader.iter_time_chunks(str(tmp_path / "stack.tif"), 3))
    assert [b[2].shape for b in got] == [(3, 2, 16, 20), (1, 2, 16, 20)] and got[1][0] == [3]
    np.testing.assert_array_equal(np.concatenate([b[2] for b in got]), hs)
    with pytest.raises(FileNotFoundError):
        list(reader.iter_time_chunks(str(tmp_path / "nothing_(time).tif"), 2))


def test_save_load_in_parts(tmp_path):
    """mg.save splits along the marker axis when asked to (or when a variable exceeds NetCDF-3's 4 GiB) and
    mg.load reassembles the parts, chips included (file.py:6-17)."""
    import magnify_amd as mg

    ds = _bead_result_dataset()
    mg.save(tmp_path / "beads.nc", ds, shard_bytes=300)  # roi is 3 marks x 256 B: one mark per part
    import glob

    parts = sorted(glob.glob(str(tmp_path / "beads.nc.part*")))
    assert len(parts) == 3 and not (tmp_path / "beads.nc").exists()
    back = mg.load(tmp_path / "beads.nc")
    assert set(back.data_vars) == {"image", "roi"} and back.attrs["name"] == "assay0"
    for k in ("image", "roi"):
        np.testing.assert_array_equal(back[k].values, ds[k].values)
        assert back[k].dims == ds[k].dims and back[k].dtype == ds[k].dtype
    for k in ("fg", "bg", "x", "y", "valid", "channel"):
        np.testing.assert_array_equal(back.coords[k].values, ds.coords[k].values)
    one = mg.load(parts[1])  # every part is a complete file of its own
    np.testing.assert_array_equal(one["roi"].values, ds["roi"].values[1:2])
    assert "image" not in one.data_vars and int(one.attrs["mg_part"]) == 1
    with pytest.raises(FileNotFoundError):
        mg.load(tmp_path / "missing.nc")
    import os

    os.remove(parts[2])
    with pytest.raises(ValueError):
        mg.load(tmp_path / "beads.nc")


def test_save_load_beyond_4_gib(tmp_path):
    """A roi variable of more than 4 GiB (C4's is 9.7 GB; NetCDF-3 holds 4 GiB per variable): saved in parts of
    2 GiB, loaded back identical.  The source is a sparse memory-mapped file with a few marked pixels."""
    import magnify_amd as mg

    m, c, L = 56000, 4, 100  # 56000 x 4 x 1 x 100 x 100 uint16 = 4.48 GB
    src = np.lib.format.open_memmap(tmp_path / "roi.npy", mode="w+", dtype=np.uint16, shape=(m, c, 1, L, L))
    marks = [0, 1, 26843, 26844, 26845, 55999]  # both sides of the part boundaries
    for k in marks:
        src[k, k % c, 0, k % L, (7 * k) % L] = 1 + k % 60000
    ds = mg.Dataset(attrs={"name": "big"})
    ds["roi"] = mg.DataArray(src, ("mark", "channel", "time", "roi_y", "roi_x"))
    ds.coords["x"] = mg.DataArray(np.arange(m, dtype=np.float64)[:, None], ("mark", "time"), name="x")
    assert src.nbytes > 2**32
    mg.save(tmp_path / "big.nc", ds)
    import glob

    parts = sorted(glob.glob(str(tmp_path / "big.nc.part*")))
    assert len(parts) == 3
    del ds
    back = mg.load(tmp_path / "big.nc")
    roi = back["roi"].values
    assert roi.shape == (m, c, 1, L, L) and roi.dtype == np.uint16
    np.testing.assert_array_equal(back.coords["x"].values[:, 0], np.arange(m))
    for k in marks:
        assert roi[k, k % c, 0, k % L, (7 * k) % L] == 1 + k % 60000
    assert int(roi.sum(dtype=np.int64)) == sum(1 + k % 60000 for k in marks)


def test_filter_nonround_against_contour_oracle():
    """The product's outer-border lengths (Moore tracing per labelled component, nested components skipped) against
    the oracle's independent restatement of what the reference calls (filter.py:51-52): Suzuki-Abe border following
    with RETR_EXTERNAL + CHAIN_APPROX_SIMPLE + arcLength -- on disks, random blobs, thin and diagonal structures,
    holes, islands inside holes, masks touching the frame."""
    from magnify_amd.filter import mask_perimeter
    from oracle import ref_contours as rc

    rng = np.random.default_rng(77)
    masks = []
    yy, xx = np.mgrid[0:40, 0:48]
    for _ in range(40):  # unions of a few disks / rectangles, with bites taken out
        m = np.zeros((40, 48), bool)
        for _ in range(rng.integers(1, 5)):
            cy, cx, r = rng.integers(0, 40), rng.integers(0, 48), rng.integers(1, 12)
            if rng.random() < 0.6:
                m |= (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
            else:
                m[max(cy - r, 0):cy + r, max(cx - r // 2, 0):cx + r // 2 + 1] = True
        for _ in range(rng.integers(0, 3)):
            cy, cx, r = rng.integers(0, 40), rng.integers(0, 48), rng.integers(1, 6)
            m &= ~((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
        masks.append(m)
    for p in (0.1, 0.3, 0.5, 0.7, 0.9):  # salt and pepper: single pixels, diagonal contacts, tiny holes
        masks += [rng.random((24, 24)) < p for _ in range(12)]
    ring = np.ones((15, 15), bool)
    ring[3:12, 3:12] = False
    island = ring.copy()
    island[6:9, 6:9] = True            # a component inside the hole of another: skipped by RETR_EXTERNAL
    nested = island.copy()
    nested[7, 7] = False               # ... with a hole of its own
    masks += [ring, island, nested, np.ones((6, 9), bool), np.zeros((5, 5), bool), np.eye(9, dtype=bool),
              np.fliplr(np.eye(9, dtype=bool)), np.triu(np.ones((12, 12), bool))]
    n_nested = 0
    for m in masks:
        want = rc.mask_perimeter(m)
        assert mask_perimeter(m) == pytest.approx(want, rel=1e-12, abs=1e-12)
        n_nested += len(rc.external_contours(m)) < scipy_components(m)
    assert rc.mask_perimeter(island) == rc.mask_perimeter(ring) == 56.0
    assert n_nested >= 3  # the set does hold masks where not every component is an external one
    # and the component's verdicts follow the oracle's roundness
    fg = np.stack(masks[:40])[:, None]
    xp = mg.Dataset({"roi": mg.DataArray(np.zeros((40, 1, 1, 40, 48), np.uint16), ("mark", "channel", "time", "roi_y", "roi_x"))},
                    coords={"fg": (("mark", "time", "roi_y", "roi_x"), fg), "valid": (("mark", "time"), np.ones((40, 1), bool))})
    got = mg.components.get("filter_nonround")(min_roundness=0.6)(xp).valid.values[:, 0]
    want = [(rc.roundness(m) or 0.0) > 0.6 for m in masks[:40]]
    assert got.tolist() == want and 5 < sum(want) < 35


def scipy_components(mask):
    import scipy.ndimage

    return scipy.ndimage.label(mask, structure=np.ones((3, 3), bool))[1]


def test_roi_chunk_policy():
    """The reference's dask chunking of roi / fg / bg (find.py:63-88, 185-201, 506-531) as numbers: markers per chunk so
    that a chunk holds >= 1e6 pixels with every channel and timepoint of a marker together, bounded by the marker
    count (chips: by the ROW count, as the reference does); carried by the variables and honoured by mg.save's parts."""
    import math

    import magnify_amd as mg
    from magnify_amd.utils import roi_mark_chunk

    assert roi_mark_chunk(2000, 4, 1, 100) == math.ceil(1e6 / (100 * 100 * 4)) == 25   # C2 / C4
    assert roi_mark_chunk(2000, 4, 64, 100) == 1                                       # mode R: a marker is 2.56 Mpx
    assert roi_mark_chunk(5, 1, 1, 20) == 5 and roi_mark_chunk(0, 1, 1, 20) == 0       # never more than there are
    assert roi_mark_chunk(28, 2, 3, 72) == min(math.ceil(1e6 / (72 * 72 * 6)), 28) == 28  # chip: bounded by 28 rows
    a = mg.DataArray(np.zeros((7, 2, 1, 4, 4), np.uint16), ("mark", "channel", "time", "roi_y", "roi_x"))
    assert a.chunks is None and a.chunksizes == {}
    b = a.chunk({"mark": 3})
    assert b.chunks == ((3, 3, 1), (2,), (1,), (4,), (4,)) and b.chunksizes["mark"] == (3, 3, 1)
    assert a.chunks is None                                             # chunk() returns a new object
    assert b.transpose("channel", "mark", ...).chunksizes["mark"] == (3, 3, 1)  # the policy follows the variable
    assert b.isel(mark=[0, 1]).chunksizes["mark"] == (2,)


def test_save_replaces_what_an_earlier_save_left(tmp_path):
    """Parts of an earlier, larger save (or the unsharded file) do not survive a new save under the same name; a file
    AND parts of it are refused by load; parts end on chunk boundaries of the variables' chunk policy."""
    import glob

    import magnify_amd as mg

    ds = _bead_result_dataset()
    mg.save(tmp_path / "b.nc", ds, shard_bytes=300)           # three parts
    assert len(glob.glob(str(tmp_path / "b.nc.part*"))) == 3
    mg.save(tmp_path / "b.nc", ds, shard_bytes=600)           # two parts now: the third must be gone
    assert len(glob.glob(str(tmp_path / "b.nc.part*"))) == 2
    np.testing.assert_array_equal(mg.load(tmp_path / "b.nc")["roi"].values, ds["roi"].values)
    mg.save(tmp_path / "b.nc", ds)                            # unsharded: no parts left
    assert glob.glob(str(tmp_path / "b.nc.part*")) == [] and (tmp_path / "b.nc").exists()
    mg.save(tmp_path / "b.nc", ds, shard_bytes=300)           # and back: the single file is gone
    assert not (tmp_path / "b.nc").exists()
    assert glob.glob(str(tmp_path / "*.tmp*")) == []          # written under temporary names, renamed
    mg.save(tmp_path / "c.nc", ds)
    import shutil

    shutil.copy(glob.glob(str(tmp_path / "b.nc.part000"))[0], tmp_path / "c.nc.part000")
    with pytest.raises(ValueError, match="both"):
        mg.load(tmp_path / "c.nc")
    # chunk-aligned parts: 7 marks, policy 2 per chunk, room for 3 marks per part -> parts of 2, 2, 2, 1
    rng = np.random.default_rng(4)
    big = mg.Dataset()
    big["roi"] = mg.DataArray(rng.integers(0, 9, (7, 1, 1, 8, 8)).astype(np.uint16), ("mark", "channel", "time", "roi_y", "roi_x")).chunk(mark=2)
    mg.save(tmp_path / "d.nc", big, shard_bytes=3 * 128)
    sizes = [mg.load(p)["roi"].shape[0] for p in sorted(glob.glob(str(tmp_path / "d.nc.part*")))]
    assert sizes == [2, 2, 2, 1]
    np.testing.assert_array_equal(mg.load(tmp_path / "d.nc")["roi"].values, big["roi"].values)


def test_sink_assay_dataset_roundtrip(tmp_path):
    """What a stream sink makes of one timepoint (magnify_amd/sink.py): the reference's bead schema (find.py:503-555)
    plus the fused reductions, saved and loaded back equal."""
    import magnify_amd as mg
    from magnify_amd.sink import _Sink

    rng = np.random.default_rng(2)
    m, c, L = 5, 3, 10
    beads = np.column_stack([rng.integers(0, 500, m), rng.integers(0, 500, m), rng.integers(5, 12, m)]).astype(np.int32)
    arrays = {"roi": rng.integers(0, 60000, (m, c, 1, L, L)).astype(np.uint16), "fg": (rng.random((m, L, L)) > 0.6).astype(np.uint8),
              "bg": (rng.random((m, L, L)) > 0.5).astype(np.uint8), "sums": rng.integers(0, 10**6, (m, c, 1, 2)).astype(np.float64),
              "counts": rng.integers(0, 99, (m, 2)).astype(np.int32)}
    ds = _Sink.assay_dataset(beads, arrays, time_label=1700000000, channels=["a", "b", "c"])
    assert ds.roi.dims == ("mark", "channel", "time", "roi_y", "roi_x") and ds.fg.dims == ("mark", "time", "roi_y", "roi_x")
    assert ds.fg.dtype == np.bool_ and ds.x.dims == ("mark", "time") and ds.valid.values.all()
    np.testing.assert_array_equal(ds.x.values[:, 0], beads[:, 1])  # x = column, y = row (find.py:543-550)
    np.testing.assert_array_equal(ds.y.values[:, 0], beads[:, 0])
    mg.save(tmp_path / "t.nc", ds)
    back = mg.load(tmp_path / "t.nc")
    for k in ("roi", "fg_sum", "bg_sum", "fg_count", "radius"):
        np.testing.assert_array_equal(back[k].values, ds[k].values)
    for k in ("fg", "bg", "x", "y", "valid"):
        np.testing.assert_array_equal(back.coords[k].values, ds.coords[k].values)
    assert list(back.channel.values) == ["a", "b", "c"] and int(back.time.values[0]) == 1700000000
    empty = _Sink.assay_dataset(np.empty((0, 3), np.int32), {k: v[:0] for k, v in arrays.items()})
    assert empty.roi.shape == (0, c, 1, L, L) and empty.x.shape == (0, 1)


def test_squeeze_of_several_dimensions_equals_one_at_a_time():
    """DataArray.squeeze (one reshape for all listed dimensions, restore_format's use) against dropping the dimensions
    one at a time with isel: data, dims and the coords that lose a dimension (they stay as lower-dimensional coords)."""
    rng = np.random.default_rng(3)
    data = rng.integers(0, 100, (1, 3, 1, 4, 1))
    coords = {"b": mg_dataarray(np.array([10, 20, 30]), ("b",), "b"),
              "a": mg_dataarray(np.array([7]), ("a",), "a"),
              "ab": mg_dataarray(rng.random((1, 3)), ("a", "b"), "ab"),
              "ce": mg_dataarray(np.array([[5.5]]), ("c", "e"), "ce")}
    v = mg.DataArray(data, ("a", "b", "c", "d", "e"), coords, "v", {"k": 1})
    for dims in (["a"], ["a", "c"], ["e", "a", "c"], None):
        got = v.squeeze(dims)
        want = v
        for d in (dims if dims is not None else ["a", "c", "e"]):
            want = want.isel({d: 0})
        assert got.dims == want.dims and got.name == "v" and got.attrs == {"k": 1}
        np.testing.assert_array_equal(got.values, want.values)
        assert set(got.coords) == set(want.coords)
        for k in got.coords:
            assert got.coords[k].dims == want.coords[k].dims, k
            np.testing.assert_array_equal(got.coords[k].values, want.coords[k].values)
    with pytest.raises(ValueError):
        v.squeeze("b")
    assert v.squeeze([]) is v


def test_process_stream_numbers_a_shard_globally(tmp_path, monkeypatch):
    """stack.process_stream(first_timepoint=lo) (config C5 across ranks): assay indices, seeds and the first_timepoint
    of every chunk are those of the unsharded run; the reader's ring is checked against the prefetch depth; the sink is
    closed however the stream ends.  (StackProcessor needs a GPU: a stand-in records what it is called with.)"""
    import torch

    from magnify_amd import stack

    calls = []

    class Fake:
        def __init__(self, t, c, h, w, **kw):
            self.shape, self.pool_tag = (t, c, h, w), ""

        def __call__(self, block, flatfield, darkfield, seed=0, want_roi=False):
            calls.append((int(block[0, 0, 0, 0]), seed, self.pool_tag))
            return {"beads": [np.zeros((0, 3), np.int32)] * block.shape[0], "sums": None, "counts": None}

    monkeypatch.setattr(stack, "StackProcessor", Fake)
    series = np.arange(10, dtype=np.uint16).reshape(10, 1, 1, 1) * np.ones((10, 2, 4, 4), dtype=np.uint16)

    class Chunks:  # an iterator that carries what reader._ChunkIter carries
        def __init__(self, lo, hi, step, ring=None):
            self.first_timepoint, self.ring, self._it, self.closed = lo, ring, iter(range(lo, hi, step)), False
            self.hi, self.step = hi, step

        def __iter__(self):
            return self

        def __next__(self):
            t = next(self._it)
            return torch.from_numpy(series[t: min(t + self.step, self.hi)].copy())

        def close(self):
            self.closed = True

    class Sink:
        closed = 0

        def __call__(self, out):
            return None

        def close(self):
            Sink.closed += 1

    whole = [o["first_timepoint"] for o in stack.process_stream(Chunks(0, 10, 3), seed=11)]
    whole_calls, calls[:] = list(calls), []
    assert whole == [0, 3, 6, 9] and [c[1] for c in whole_calls] == [11 + 1000003 * t for t in whole]
    part = [o["first_timepoint"] for o in stack.process_stream(Chunks(6, 10, 3), seed=11, sink=Sink())]
    assert part == [6, 9] and calls == [(6, 11 + 1000003 * 6, "#0"), (9, 11 + 1000003 * 9, "#1")] and Sink.closed == 1
    assert [c[:2] for c in calls] == [c[:2] for c in whole_calls[2:]]  # the shard's calls ARE the whole run's
    # an explicit first_timepoint wins over the iterator's
    assert [o["first_timepoint"] for o in stack.process_stream(Chunks(6, 10, 3), first_timepoint=100)] == [100, 103]
    # a ring smaller than what the stream holds is refused before anything is read
    with pytest.raises(ValueError, match="ring"):
        next(stack.process_stream(Chunks(0, 10, 3, ring=3), prefetch=2))
    next(stack.process_stream(Chunks(0, 10, 3, ring=4), prefetch=2))
    # the consumer walks away after one chunk: the sink is closed, the reader's iterator too
    src, before = Chunks(0, 10, 1), Sink.closed
    gen = stack.process_stream(src, sink=Sink(), prefetch=1)
    next(gen)
    gen.close()
    assert Sink.closed == before + 1
    for _ in range(100):
        if src.closed:
            break
        import time
        time.sleep(0.05)
    assert src.closed
    # an error inside a chunk: the sink is still closed
    before = Sink.closed

    def bad():
        yield torch.from_numpy(series[:2].copy())
        raise OSError("disk gone")

    with pytest.raises(OSError):
        list(stack.process_stream(bad(), sink=Sink()))
    assert Sink.closed == before + 1


def test_reader_ring_belongs_to_one_iterator(tmp_path):
    """blocks of iter_time_chunks(ring=...) are owned by ONE live iterator -- two iterators over same-shape
    series never share a block; a block comes back after ``ring`` chunks, not earlier; the pool takes the blocks back
    when an iterator ends (also early)."""
    from PIL import Image

    from magnify_amd import reader

    rng = np.random.default_rng(0)
    data = rng.integers(0, 60000, (6, 2, 16, 24)).astype(np.uint16)
    for t in range(6):
        for c in range(2):
            Image.fromarray(data[t, c]).save(tmp_path / f"s_c{c}_2024010{t + 1}.tif")
    pattern = str(tmp_path / "s_c(channel)_(time|%Y%m%d).tif")
    reader.release_pinned()
    a, b = reader.iter_time_chunks(pattern, 1, ring=3), reader.iter_time_chunks(pattern, 1, ring=3)
    seen_a, seen_b = [], []
    for k in range(6):
        ba, bb = next(a)[2], next(b)[2]
        np.testing.assert_array_equal(ba, data[k: k + 1])
        np.testing.assert_array_equal(bb, data[k: k + 1])
        seen_a.append(ba.ctypes.data)
        seen_b.append(bb.ctypes.data)
    assert not set(seen_a) & set(seen_b)
    assert len(set(seen_a)) == 3 and seen_a[:3] == seen_a[3:]  # overwritten after `ring` further chunks, not before
    with pytest.raises(StopIteration):
        next(a)
    c = reader.iter_time_chunks(pattern, 1, ring=3)  # a's blocks are free again: c pins nothing new
    assert next(c)[2].ctypes.data in set(seen_a)
    c.close()
    b.close()
    assert sum(len(v) for v in reader._POOL._free.values()) == 6
    reader.release_pinned()
    # without a ring every chunk is a fresh array (list() of the iterator holds distinct blocks)
    blocks = [blk for _, _, blk in reader.iter_time_chunks(pattern, 2)]
    np.testing.assert_array_equal(np.concatenate(blocks), data)


def test_host_read_runs(tmp_path):
    """mg_host_read_runs (the native reader threads of the streamed ingest): byte runs of several files land where
    they should, whatever the thread count; the end of a file and a bad descriptor are reported, not ignored."""
    import ctypes

    from magnify_amd import _native

    rng = np.random.default_rng(1)
    blobs = [rng.integers(0, 256, n, dtype=np.uint8) for n in (100_000, 20_000_000, 7)]
    fds = []
    for i, b in enumerate(blobs):
        (tmp_path / f"f{i}.bin").write_bytes(b.tobytes())
        fds.append(os.open(tmp_path / f"f{i}.bin", os.O_RDONLY))
    runs = [(0, 0, 100_000), (1, 5, 19_999_990), (2, 0, 7), (1, 12_345, 1_000_001), (0, 99_999, 1), (2, 3, 0)]
    for threads in (1, 3, 16):
        outs = [np.zeros(n, dtype=np.uint8) for _, _, n in runs]
        fd = np.array([fds[f] for f, _, _ in runs], dtype=np.int32)
        off = np.array([o for _, o, _ in runs], dtype=np.int64)
        nb = np.array([n for _, _, n in runs], dtype=np.int64)
        dst = np.array([o.ctypes.data for o in outs], dtype=np.uint64)
        assert _native.lib().mg_host_read_runs(fd.ctypes.data, off.ctypes.data, nb.ctypes.data, dst.ctypes.data, len(runs),
                                               threads, None) == 0
        for (f, o, n), got in zip(runs, outs):
            np.testing.assert_array_equal(got, blobs[f][o: o + n])
    failed = (ctypes.c_int64 * 2)(-1, -1)
    nb[2] = 8  # one byte beyond the end of the 7-byte file
    outs[2] = np.zeros(8, dtype=np.uint8)
    dst[2] = outs[2].ctypes.data
    assert _native.lib().mg_host_read_runs(fd.ctypes.data, off.ctypes.data, nb.ctypes.data, dst.ctypes.data, len(runs), 2,
                                           ctypes.addressof(failed)) == -3
    assert failed[0] == 2 and failed[1] == 0
    assert _native.lib().mg_host_read_runs(fd.ctypes.data, off.ctypes.data, nb.ctypes.data, dst.ctypes.data, len(runs), 0, None) == -1
    for f in fds:
        os.close(f)
    # through the TIFF layer: a truncated file is an error with the file's name in it
    from tiffwrite import write_tiff

    from magnify_amd import tiff

    page = rng.integers(0, 60000, (64, 64)).astype(np.uint16)
    write_tiff(tmp_path / "ok.tif", [page, page + 1])
    with tiff.TiffFile(tmp_path / "ok.tif") as tif:
        a, b = np.empty((64, 64), np.uint16), np.empty((64, 64), np.uint16)
        tiff.read_pages([(tif, 0, a), (tif, 1, b)], workers=4)
        np.testing.assert_array_equal(a, page)
        np.testing.assert_array_equal(b, page + 1)
        with pytest.raises(tiff.TiffError):
            tiff.read_pages([(tif, 0, np.empty((64, 65), np.uint16))], workers=1)


def test_lzw_megabyte_strip(tmp_path):
    """the LZW decoder's bit accumulator is cut back after every code -- a strip of a megabyte decodes in
    seconds (it was quadratic in the strip length)."""
    import time

    from PIL import Image

    from magnify_amd import tiff

    rng = np.random.default_rng(2)
    img = (rng.integers(0, 4, (1024, 1024)) * 60).astype(np.uint8)  # compressible: long codes, frequent table resets
    Image.fromarray(img).save(tmp_path / "lzw.tif", compression="tiff_lzw", tiffinfo={278: 1024})
    with tiff.TiffFile(tmp_path / "lzw.tif") as tif:
        assert tif.page(0).compression == 5 and max(tif.page(0).counts) > 100_000
        t0 = time.perf_counter()
        got = tif.asarray(0)
        assert time.perf_counter() - t0 < 60
    np.testing.assert_array_equal(got, img)


def test_save_keeps_the_old_data_until_the_new_is_written(tmp_path, monkeypatch):
    """mg.save writes everything under temporary names and removes what an earlier save left only after the
    renames -- a failure in the middle of a multi-part save leaves the old file readable and no debris."""
    import magnify_amd as mg
    from magnify_amd import file as mgfile

    old = mg.Dataset({"roi": mg.DataArray(np.arange(24, dtype=np.uint16).reshape(6, 4), ("mark", "x"))})
    target = tmp_path / "res.nc"
    mg.save(target, old)
    new = mg.Dataset({"roi": mg.DataArray(np.arange(64, dtype=np.uint16).reshape(16, 4) + 100, ("mark", "x"))})
    real, n = mgfile._write_nc, [0]

    def failing(path, ds, threads=None):
        n[0] += 1
        if n[0] == 3:
            raise OSError("No space left on device")
        real(path, ds, threads)

    monkeypatch.setattr(mgfile, "_write_nc", failing)
    with pytest.raises(OSError):
        mg.save(target, new, shard_bytes=32)  # four parts; the third fails
    monkeypatch.setattr(mgfile, "_write_nc", real)
    assert sorted(p.name for p in tmp_path.iterdir()) == ["res.nc"]
    np.testing.assert_array_equal(mg.load(target)["roi"].values, old["roi"].values)
    mg.save(target, new, shard_bytes=32)
    assert sorted(p.name for p in tmp_path.iterdir()) == [f"res.nc.part{k:03d}" for k in range(4)]
    np.testing.assert_array_equal(mg.load(target)["roi"].values, new["roi"].values)
    mg.save(target, old)  # back to one file: the parts go
    assert sorted(p.name for p in tmp_path.iterdir()) == ["res.nc"]
