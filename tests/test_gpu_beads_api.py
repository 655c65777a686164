"""End-to-end (level-2) parity: the scenarios and tolerances of the reference's own bead and
stitch tests (tests/test_beads.py, tests/test_stitch.py), driven through the drop-in API
``mg.beads`` / ``Stitcher`` of this build.  Statistical tolerances, as in the reference, because
the reference detector is unseeded."""
import numpy as np
import pytest

from oracle import ref_pipeline as rp
from synth import draw_beads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mg():
    import magnify_amd

    magnify_amd.hotpath = __import__("magnify_amd.hotpath", fromlist=["x"])
    magnify_amd.hotpath.require_gpu()
    magnify_amd.seed(1234)
    return magnify_amd


def arr(mg, data, dims, **coords):
    return mg.DataArray(data=data, dims=dims, coords=coords or None)


def test_bead_single(mg):
    # tests/test_beads.py:49-66
    xp = mg.beads(data=arr(mg, draw_beads((1024, 1024), [512, 512]), ("y", "x")), min_bead_diameter=16,
                  max_bead_diameter=24, overlap=0, num_iter=100)
    assert isinstance(xp, mg.Dataset)
    assert xp.roi.sizes["mark"] == 1
    radius = 10
    detected = np.sqrt(xp.fg.sum().values.item() / np.pi)
    assert 0.95 * radius < detected < 1.05 * radius
    assert 0.95 * 512 < xp.x.squeeze().values.item() < 1.05 * 512
    assert 0.95 * 512 < xp.y.squeeze().values.item() < 1.05 * 512


def test_beads_multiple(mg):
    # tests/test_beads.py:69-98
    pos = [[200, 200], [200, 800], [512, 512], [800, 200], [800, 800]]
    xp = mg.beads(data=arr(mg, draw_beads((1024, 1024), pos), ("y", "x")), min_bead_diameter=16, max_bead_diameter=24,
                  overlap=0, num_iter=10000)
    assert xp.roi.sizes["mark"] == 5
    radii = np.sqrt(xp.fg.sum(dim=["roi_x", "roi_y"]).values / np.pi)
    assert np.all(radii > 9) and np.all(radii < 11)


def test_beads_near_edges(mg):
    # tests/test_beads.py:101-130
    pos = [[50, 512], [974, 512], [512, 50], [512, 974]]
    xp = mg.beads(data=arr(mg, draw_beads((1024, 1024), pos), ("y", "x")), min_bead_diameter=16, max_bead_diameter=24,
                  overlap=0, num_iter=10000)
    assert xp.roi.sizes["mark"] == 4
    x, y = xp.x.squeeze().values, xp.y.squeeze().values
    assert np.any(y < 100) and np.any(y > 900) and np.any(x < 100) and np.any(x > 900)


def test_beads_close_stay_separate(mg):
    # tests/test_beads.py:160-188: two beads 30 px apart are both found
    pos = [[500, 500], [500, 530]]
    xp = mg.beads(data=arr(mg, draw_beads((1024, 1024), pos), ("y", "x")), min_bead_diameter=16, max_bead_diameter=24,
                  overlap=0, num_iter=10000)
    assert xp.roi.sizes["mark"] == 2
    xs = np.sort(xp.x.squeeze().values)
    assert abs(xs[0] - 500) < 5 and abs(xs[1] - 530) < 5


def test_beads_different_sizes(mg):
    # tests/test_beads.py:191-216
    pos = [[300, 300], [300, 700], [700, 500]]
    diam = [16, 24, 32]
    xp = mg.beads(data=arr(mg, draw_beads((1024, 1024), pos, np.array(diam)), ("y", "x")), min_bead_diameter=12,
                  max_bead_diameter=40, overlap=0, num_iter=20000)
    assert xp.roi.sizes["mark"] == 3
    radii = np.sort(np.sqrt(xp.fg.sum(dim=["roi_x", "roi_y"]).values / np.pi))
    for got, want in zip(radii, [8, 12, 16]):
        assert 0.8 * want < got < 1.2 * want


def test_beads_empty_image(mg):
    # tests/test_beads.py:219-232
    xp = mg.beads(data=arr(mg, np.zeros((512, 512), dtype=np.uint16), ("y", "x")), min_bead_diameter=16,
                  max_bead_diameter=24, overlap=0, num_iter=1000)
    assert xp.roi.sizes["mark"] == 0


def test_beads_float32_input(mg):
    # tests/test_beads.py:235-247
    img = draw_beads((1024, 1024), [512, 512]).astype(np.float32)
    xp = mg.beads(data=arr(mg, img, ("y", "x")), min_bead_diameter=16, max_bead_diameter=24, overlap=0, num_iter=1000)
    assert xp.roi.sizes["mark"] == 1
    assert xp.roi.dtype == np.float32


def test_beads_output_structure(mg):
    # tests/test_beads.py:250-274
    xp = mg.beads(data=arr(mg, draw_beads((1024, 1024), [512, 512]), ("y", "x")), min_bead_diameter=16,
                  max_bead_diameter=24, overlap=0, num_iter=1000)
    for name in ("x", "y", "fg", "bg"):
        assert name in xp.coords
    assert "roi" in xp.data_vars
    assert set(xp.roi.dims) == {"mark", "roi_x", "roi_y"}
    assert "tile" not in xp.data_vars and "image" in xp.data_vars
    assert xp.roi.shape[-2:] == (48, 48)  # roi_length = 2 * max_bead_diameter (find.py:467)
    assert "__original_tile_dims__" not in xp.attrs


def test_beads_multichannel_dedup_and_time(mg):
    # tests/test_beads.py:394-430 (cross-channel de-duplication) + (channel, time, y, x) layout
    a = draw_beads((512, 512), [[150, 150], [350, 350]])
    b = draw_beads((512, 512), [[150, 150], [150, 350]])  # one bead shared with channel a
    data = np.stack([np.stack([a, a]), np.stack([b, b])])  # (channel, time, y, x)
    xp = mg.beads(data=arr(mg, data, ("channel", "time", "y", "x"), channel=["red", "green"]), min_bead_diameter=16,
                  max_bead_diameter=24, overlap=0, num_iter=5000)
    assert xp.roi.sizes["mark"] == 3
    assert xp.roi.dims == ("mark", "channel", "time", "roi_y", "roi_x")
    assert xp.fg.dims == ("mark", "time", "roi_y", "roi_x")
    np.testing.assert_array_equal(xp.x.values[:, 0], xp.x.values[:, 1])  # geometry replicated over time
    xq = mg.beads(data=arr(mg, data, ("channel", "time", "y", "x"), channel=["red", "green"]), min_bead_diameter=16,
                  max_bead_diameter=24, overlap=0, num_iter=5000, search_channel="green")
    assert xq.roi.sizes["mark"] == 2


def test_beads_roi_contents_and_reductions(mg):
    pos = [[200, 300], [400, 100]]
    img = draw_beads((512, 512), pos, 20, [1000, 3000])
    pipe = mg.beads_pipe(min_bead_diameter=16, max_bead_diameter=24, overlap=0, num_iter=5000)
    pipe.remove_pipe("restore_format")
    xp = pipe(arr(mg, img, ("y", "x")))
    assert xp.roi.sizes["mark"] == 2
    roi, fg, bg = xp.roi.values, xp.fg.values, xp.bg.values
    for i in range(2):
        x, y = int(xp.x.values[i, 0]), int(xp.y.values[i, 0])
        top, bottom, left, right = mg.utils.bounding_box(x, y, 48, 512, 512)
        np.testing.assert_array_equal(roi[i, 0, 0], img[top:bottom, left:right])
        assert fg[i, 0].sum() > 250 and not (fg[i, 0] & bg[i, 0]).any()
    s = mg.reduce.masked_sum(xp, "fg").values
    np.testing.assert_array_equal(s[:, 0, 0], (roi[:, 0, 0] * fg[:, 0]).sum(axis=(-1, -2)))
    mean = mg.reduce.masked_mean(xp, "fg").values[:, 0, 0]
    assert set(np.round(mean).astype(int).tolist()) == {1000, 3000}
    med = mg.reduce.masked_median(xp, "bg").values
    assert (med == 0).all()
    np.testing.assert_array_equal(mg.reduce.counts(xp, "fg").values[:, 0], fg[:, 0].sum(axis=(-1, -2)))
    # without the finder's cached reductions (any Dataset with roi / fg / bg): the masked-sum kernel, also for masks
    # that differ from timepoint to timepoint
    xp._cache.pop("roi_sums"), xp._cache.pop("roi_counts")
    np.testing.assert_array_equal(mg.reduce.masked_sum(xp, "bg").values[:, 0, 0], (roi[:, 0, 0].astype(np.int64) * bg[:, 0]).sum(axis=(-1, -2)))
    np.testing.assert_array_equal(mg.reduce.counts(xp, "bg").values[:, 0], bg[:, 0].sum(axis=(-1, -2)))
    rng = np.random.default_rng(0)
    roi3 = rng.integers(0, 60000, (3, 2, 3, 12, 12)).astype(np.uint16)
    fg3 = rng.random((3, 3, 12, 12)) > 0.5
    ds = mg.Dataset({"roi": mg.DataArray(roi3, ("mark", "channel", "time", "roi_y", "roi_x"))},
                    coords={"fg": (("mark", "time", "roi_y", "roi_x"), fg3)})
    np.testing.assert_array_equal(mg.reduce.masked_sum(ds, "fg").values, (roi3.astype(np.int64) * fg3[:, None]).sum(axis=(-1, -2)))
    np.testing.assert_array_equal(mg.reduce.counts(ds, "fg").values, fg3.sum(axis=(-1, -2)))
    # user-side algebra through the container (README.md:21-22)
    np.testing.assert_allclose(xp.roi.where(xp.fg).mean(dim=["roi_x", "roi_y"]).values[:, 0, 0], mean)


def test_flatfield_in_pipeline_matches_oracle(mg):
    from oracle import ref_pipeline as rp
    from synth import noisy_bead_image, vignette

    img, _ = noisy_bead_image(3, (256, 256), 5)
    tiles = np.stack([img, img[::-1].copy()]).reshape(1, 1, 1, 2, 256, 256)
    flat = vignette((256, 256))
    pipe = mg.Pipeline("read")
    pipe.add_pipe("standardize_format")
    pipe.add_pipe("flatfield_correct", flatfield=flat, darkfield=100.0)
    pipe.add_pipe("stitch", overlap=16)
    xp = pipe(mg.DataArray(tiles, ("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x")))
    want = rp.stitch(rp.flatfield_correct(tiles, flat, 100.0), 16)
    np.testing.assert_array_equal(xp.image.values, want)
    # the lazily corrected tiles materialise to the same values
    np.testing.assert_array_equal(xp.tile.values, rp.flatfield_correct(tiles, flat, 100.0))


def test_flatfield_from_tiff_files(mg, tmp_path):
    """flatfield / darkfield given as paths of TIFF files, as the reference reads them with tifffile
    (preprocess.py:64-81): same result as the arrays themselves."""
    from PIL import Image

    from oracle import ref_pipeline as rp
    from synth import noisy_bead_image, vignette

    img, _ = noisy_bead_image(4, (200, 240), 4)
    tiles = img.reshape(1, 1, 1, 1, 200, 240)
    flat = vignette((200, 240))  # float32
    dark = (90 + (np.arange(200 * 240).reshape(200, 240) % 7)).astype(np.uint16)
    Image.fromarray(flat).save(tmp_path / "flat.tif")
    Image.fromarray(dark).save(tmp_path / "dark.tif")
    pipe = mg.Pipeline("read")
    pipe.add_pipe("standardize_format")
    pipe.add_pipe("flatfield_correct", flatfield=str(tmp_path / "flat.tif"), darkfield=tmp_path / "dark.tif")
    pipe.add_pipe("stitch", overlap=0)
    xp = pipe(mg.DataArray(tiles, ("channel", "time", "tile_row", "tile_col", "tile_y", "tile_x")))
    np.testing.assert_array_equal(xp.image.values, rp.stitch(rp.flatfield_correct(tiles, flat, dark), 0))
    with pytest.raises(FileNotFoundError):
        mg.components.get("flatfield_correct")(flatfield=str(tmp_path / "nope.tif"))(xp)


def test_stitcher_component(mg):
    # tests/test_stitch.py through the component object
    from magnify_amd.stitch import Stitcher

    rng = np.random.default_rng(0)
    dims = ["channel", "time", "tile_row", "tile_col", "tile_y", "tile_x"]
    t = rng.random((2, 3, 2, 2, 25, 25))
    ds = mg.Dataset({"tile": mg.DataArray(t, dims)}, coords={"channel": ["red", "green"], "time": [0, 1, 2]})
    res = Stitcher(overlap=8)(ds)
    assert "image" in res.data_vars and res.image.dims == ("channel", "time", "im_y", "im_x")
    assert len(res.channel) == 2 and len(res.time) == 3
    assert res.sizes["im_y"] == 2 * (25 - 8)
    with pytest.raises(ValueError):
        Stitcher(overlap=-5)
    with pytest.raises(AttributeError):
        Stitcher(overlap=10)(mg.Dataset({"other": mg.DataArray([1, 2, 3], ("x",))}))
    with pytest.raises(ValueError):
        Stitcher(overlap=100)(mg.Dataset({"tile": mg.DataArray(rng.random((1, 1, 2, 2, 50, 50)), dims)}))


def test_mrbles_front_half(mg, tmp_path):
    """identify.py:50-90 through mg.mrbles: fg mean - bg median per channel, least-squares lanthanide
    volumes against the spectra, ratios to the reference lanthanide; and the reference's ValueErrors."""
    rng = np.random.default_rng(5)
    spectra = np.array([[1.0, 0.2, 0.0], [0.1, 1.0, 0.3], [0.0, 0.25, 1.0]])  # (lanthanide, channel)
    lns, chans = ["eu", "dy", "sm"], ["c620", "c572", "c600"]
    pos = [[120, 120], [120, 360], [360, 120], [360, 360], [240, 240]]
    vols = rng.uniform(200, 900, size=(len(pos), 3))
    data = np.zeros((3, 480, 480), dtype=np.uint16)
    for c in range(3):
        inten = (vols @ spectra[:, c]).round().astype(int)
        data[c] = draw_beads((480, 480), pos, 20, inten) + np.uint16(100)  # flat background, like the reference's tests
    sp_csv, codes_csv = tmp_path / "spectra.csv", tmp_path / "codes.csv"
    # the file lists dy first: the reference lanthanide must still come out first
    sp_csv.write_text("name," + ",".join(chans) + "\n" + "dy," + ",".join(map(str, spectra[1])) + "\n" +
                      "eu," + ",".join(map(str, spectra[0])) + "\n" + "sm," + ",".join(map(str, spectra[2])) + "\n")
    codes_csv.write_text("name,eu,dy,sm\ncode0,1,0.5,0.2\ncode1,1,1.0,0.4\n")
    pipe = mg.mrbles_pipe(spectra=str(sp_csv), codes=str(codes_csv), min_bead_diameter=16, max_bead_diameter=24,
                          overlap=0, num_iter=20000)
    pipe.remove_pipe("restore_format")
    xp = pipe(arr(mg, data, ("channel", "y", "x"), channel=chans))
    assert list(xp.ln.values) == ["eu", "dy", "sm"]
    assert xp.tag.dims == ("mark",) and set(xp.tag.values.tolist()) <= {"code0", "code1", "outlier"}  # identify.py:224-231
    m = xp.roi.sizes["mark"]
    assert m == len(pos) and xp.ln_vol.shape == (m, 3) and xp.ln_ratio.shape == (m, 3)
    # oracle: the same expression on the returned arrays
    roi = xp.roi.transpose("mark", "channel", "time", "roi_y", "roi_x").values
    fg = xp.fg.transpose("mark", "time", "roi_y", "roi_x").values
    bg = xp.bg.transpose("mark", "time", "roi_y", "roi_x").values
    red = rp.roi_reduce(roi, fg, bg)
    inten = (red["fg_mean"] - red["bg_median"])[:, :, 0]
    sp = np.stack([spectra[0], spectra[1], spectra[2]])  # eu, dy, sm
    want = np.linalg.lstsq(sp.T, inten.T, rcond=None)[0].T
    np.testing.assert_array_equal(xp.ln_vol.values, want)
    np.testing.assert_array_equal(xp.ln_ratio.values, want / want[:, 0:1])
    # beads are found in some order: match by position, volumes recovered to a few percent
    got = {(int(round(y / 120)), int(round(x / 120))): v for y, x, v in zip(xp.y.values[:, 0], xp.x.values[:, 0], want)}
    for p, v in zip(pos, vols):
        np.testing.assert_allclose(got[(p[0] // 120, p[1] // 120)], v, rtol=0.05)
    with pytest.raises(ValueError):
        mg.mrbles(arr(mg, data, ("channel", "y", "x"), channel=chans), spectra=str(sp_csv), codes=str(codes_csv),
                  reference="tm", min_bead_diameter=16, max_bead_diameter=24, overlap=0, num_iter=2000)
    bad = tmp_path / "bad.csv"
    bad.write_text("name,eu,dy\ncode0,1,0.5\n")
    with pytest.raises(ValueError):
        mg.mrbles(arr(mg, data, ("channel", "y", "x"), channel=chans), spectra=str(sp_csv), codes=str(bad),
                  min_bead_diameter=16, max_bead_diameter=24, overlap=0, num_iter=2000)


# ---- the remaining scenarios of the reference's tests/test_beads.py, one to one ---------------------------


def _positions_100(xp, n):
    return {(round(xp.y[i].values.item() / 100) * 100, round(xp.x[i].values.item() / 100) * 100) for i in range(n)}


def test_beads_varying_sizes(mg):
    # tests/test_beads.py:131-157
    pos = [[300, 300], [300, 700], [700, 300], [700, 700]]
    img = draw_beads((1024, 1024), pos, np.array([16, 20, 24, 28]))
    xp = mg.beads(data=arr(mg, img, ("y", "x")), min_bead_diameter=14, max_bead_diameter=32, overlap=0, num_iter=10000)
    assert isinstance(xp, mg.Dataset) and xp.roi.sizes["mark"] == 4
    areas = xp.fg.sum(dim=["roi_x", "roi_y"]).values
    assert areas.max() / areas.min() > 1.5


def test_beads_varying_intensity(mg):
    # tests/test_beads.py:191-216
    pos = [[300, 500], [500, 500], [700, 500]]
    img = draw_beads((1024, 1024), pos, 20, [500, 1000, 2000])
    xp = mg.beads(data=arr(mg, img, ("y", "x")), min_bead_diameter=16, max_bead_diameter=24, overlap=0, num_iter=10000)
    assert xp.roi.sizes["mark"] == 3
    radii = np.sqrt(xp.fg.sum(dim=["roi_x", "roi_y"]).values / np.pi)
    assert np.all(radii > 0.85 * 10)


def test_beads_multichannel_search_single(mg):
    # tests/test_beads.py:282-324
    pos = [[300, 300], [700, 700]]
    data = np.stack([draw_beads((1024, 1024), pos), draw_beads((1024, 1024), pos)])
    xp = mg.beads(data=arr(mg, data, ("channel", "y", "x"), channel=["red", "green"]), min_bead_diameter=16,
                  max_bead_diameter=24, overlap=0, num_iter=5000, search_channel="red")
    assert xp.roi.sizes["mark"] == 2
    assert "red" in xp.channel.values and "green" in xp.channel.values
    assert _positions_100(xp, 2) == {(300, 300), (700, 700)}
    for area in xp.fg.sum(dim=["roi_x", "roi_y"]).values:
        assert 0.8 * 10 < np.sqrt(area / np.pi) < 1.2 * 10


def test_beads_multichannel_different_beads(mg):
    # tests/test_beads.py:327-365
    data = np.stack([draw_beads((1024, 1024), [[200, 200], [200, 800]]), draw_beads((1024, 1024), [[800, 200], [800, 800]])])
    xp = mg.beads(data=arr(mg, data, ("channel", "y", "x"), channel=["red", "green"]), min_bead_diameter=16,
                  max_bead_diameter=24, overlap=0, num_iter=10000, search_channel=["red", "green"])
    assert xp.roi.sizes["mark"] == 4
    assert _positions_100(xp, 4) == {(200, 200), (200, 800), (800, 200), (800, 800)}


def test_beads_multichannel_subset_only(mg):
    # tests/test_beads.py:368-391: a bead that only exists in a channel that is not searched is not found
    data = np.stack([np.zeros((1024, 1024), dtype=np.uint16), draw_beads((1024, 1024), [[512, 512]])])
    xp = mg.beads(data=arr(mg, data, ("channel", "y", "x"), channel=["red", "green"]), min_bead_diameter=16,
                  max_bead_diameter=24, overlap=0, num_iter=1000, search_channel="red")
    assert isinstance(xp, mg.Dataset) and xp.roi.sizes["mark"] == 0


def test_mrbles_front_half_on_float32_input(mg, tmp_path):
    """identify.py:76-80 works on whatever dtype the images have (tests/test_beads.py:235-247 runs the
    finder on float32).  mg.mrbles on float32 input: fg mean - bg median of time 0 from the device reductions (float32
    medians by radix select on the bit pattern) equal the oracle's on the returned arrays; also with a second
    timepoint whose pixels differ, which the result must not depend on (``assay.roi.isel(time=0)`` comes first)."""
    rng = np.random.default_rng(6)
    spectra = np.array([[1.0, 0.2], [0.1, 1.0]])  # (lanthanide, channel)
    chans = ["c620", "c572"]
    pos = [[120, 120], [120, 360], [360, 120], [360, 360]]
    vols = rng.uniform(200, 900, size=(len(pos), 2))
    data = np.zeros((2, 480, 480), dtype=np.float32)
    for c in range(2):
        inten = (vols @ spectra[:, c]).round().astype(int)
        data[c] = draw_beads((480, 480), pos, 20, inten).astype(np.float32) + np.float32(100.25)
    data += rng.normal(0, 0.5, size=data.shape).astype(np.float32)  # fractional values: the medians are no integers
    sp_csv, codes_csv = tmp_path / "spectra.csv", tmp_path / "codes.csv"
    sp_csv.write_text("name," + ",".join(chans) + "\n" + "eu," + ",".join(map(str, spectra[0])) + "\n" +
                      "dy," + ",".join(map(str, spectra[1])) + "\n")
    codes_csv.write_text("name,eu,dy\ncode0,1,0.5\ncode1,1,1.0\n")
    kw = dict(spectra=str(sp_csv), codes=str(codes_csv), min_bead_diameter=16, max_bead_diameter=24, overlap=0, num_iter=20000)
    pipe = mg.mrbles_pipe(**kw)
    pipe.remove_pipe("restore_format")
    mg.seed(77)
    xp = pipe(arr(mg, data, ("channel", "y", "x"), channel=chans))
    assert xp.roi.dtype == np.float32 and xp.roi.sizes["mark"] == len(pos)
    roi = xp.roi.transpose("mark", "channel", "time", "roi_y", "roi_x").values
    fg = xp.fg.transpose("mark", "time", "roi_y", "roi_x").values
    bg = xp.bg.transpose("mark", "time", "roi_y", "roi_x").values
    red = rp.roi_reduce(roi, fg, bg)
    from magnify_amd import reduce

    np.testing.assert_array_equal(reduce.masked_median(xp, "bg").data.cpu().numpy(), red["bg_median"])
    np.testing.assert_array_equal(reduce.masked_median(xp, "fg").data.cpu().numpy(), red["fg_median"])
    # the float sums of the masked-sum kernel are float64 accumulations of float32 pixels: the mean within 1e-12 relative
    inten = (red["fg_mean"] - red["bg_median"])[:, :, 0]
    want = np.linalg.lstsq(spectra.T, inten.T, rcond=None)[0].T
    np.testing.assert_allclose(xp.ln_vol.values, want, rtol=1e-9)
    # two timepoints, the second one different: same volumes as the first timepoint alone
    two = np.stack([data, data[:, ::-1] * np.float32(0.5)], axis=1)  # (channel, time, y, x)
    pipe = mg.mrbles_pipe(**kw)
    pipe.remove_pipe("restore_format")
    mg.seed(77)  # the same RNG stream: the same beads
    xq = pipe(arr(mg, two, ("channel", "time", "y", "x"), channel=chans))
    mg.seed(4321)
    assert xq.sizes["time"] == 2 and xq.roi.sizes["mark"] == len(pos)
    order_p = np.lexsort((xp.x.values[:, 0], xp.y.values[:, 0]))
    order_q = np.lexsort((xq.x.values[:, 0], xq.y.values[:, 0]))
    np.testing.assert_allclose(xq.ln_vol.values[order_q], xp.ln_vol.values[order_p], rtol=1e-9)
