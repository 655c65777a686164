"""Full-size (BASELINE.json shapes) checks through size-independent properties, the C3-style
stitched chip through the whole pipeline, and the single-assay (mode R) stack path."""
import numpy as np
import pytest

from oracle import ref_pipeline as rp
from synth import draw_chip, vignette

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def mg():
    import magnify_amd
    from magnify_amd import hotpath

    hotpath.require_gpu()
    magnify_amd.seed(99)
    return magnify_amd


def test_full_size_stack_properties(mg):
    """2 timepoints x 4 ch x 4096^2 with the reference-default RANSAC budget."""
    from magnify_amd.stack import StackProcessor, synthetic_stack

    T, C, S = 2, 4, 4096
    stack, truth = synthetic_stack(T, C, S, S, seed=123)
    flat_np = vignette((S, S))
    flat = torch.from_numpy(flat_np).cuda()
    proc = StackProcessor(T, C, S, S, num_iter=5_000_000, search_channels=(0,), mode="P")
    out = proc(stack, flat, 100.0, seed=7, want_roi=True)
    beads = out["beads"]
    # (a) flat-field: a random sample of rows against the oracle formula (global maxima per assay)
    img = proc.image.cpu().numpy()
    st = stack.cpu().numpy()
    for t in range(T):
        want = rp.flatfield_correct(st[t].reshape(C, 1, 1, 1, S, S), flat_np, 100.0).reshape(C, S, S)
        rows = np.random.default_rng(t).integers(0, S, 16)
        np.testing.assert_array_equal(img[t][:, rows], want[:, rows])
    # (b) recall against the drawn beads of timepoint 0 (exact positions known)
    found = beads[0][:, :2].astype(np.float64)
    d = np.sqrt(((truth[:, None, :2] - found[None]) ** 2).sum(-1)).min(axis=1)
    assert (d <= 3).mean() > 0.8  # the reference algorithm at min_roundness 0.3 misses dim beads in noise
    # (c) masks and reductions are self-consistent: counts == mask sums, sums == sum(roi * mask)
    roi, fg, bg = out["roi"], out["fg"], out["bg"]
    counts, sums = out["counts"].cpu().numpy(), out["sums"].cpu().numpy()
    np.testing.assert_array_equal(counts[:, 0], fg.sum(dim=(-1, -2), dtype=torch.int64).cpu().numpy())
    np.testing.assert_array_equal(counts[:, 1], bg.sum(dim=(-1, -2), dtype=torch.int64).cpu().numpy())
    m = roi.shape[0]
    pick = np.random.default_rng(0).integers(0, m, 200)
    r = roi.view(torch.int16)[pick].cpu().numpy().view(np.uint16).astype(np.int64)
    f = fg[pick].cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(sums[pick][:, :, 0, 0], (r[:, :, 0] * f[:, None]).sum(axis=(-1, -2)))
    assert not (fg & bg).any()
    # (d) same seed -> identical result (the whole chain is deterministic)
    out2 = proc(stack, flat, 100.0, seed=7, want_roi=False)
    for a, b in zip(beads, out2["beads"]):
        np.testing.assert_array_equal(a, b)


def test_streamed_detection_is_identical(mg):
    """Detection split over HIP streams / host threads gives exactly the single-stream result."""
    from magnify_amd.stack import StackProcessor, synthetic_stack

    stack, _ = synthetic_stack(6, 2, 512, 640, seed=77)
    outs = []
    for n_streams, sub in ((1, None), (2, None), (3, None), (2, 6), (3, 6)):
        proc = StackProcessor(6, 2, 512, 640, num_iter=100000, search_channels=(0, 1), mode="P", n_streams=n_streams,
                              sub_batches=sub)
        outs.append(proc(stack, 0.9, 100.0, seed=5))
        assert proc.n_streams == n_streams
        if n_streams > 1:
            assert len(proc.ranges) == (sub or n_streams)
    # host ingest: the same stack handed over in pinned host memory, uploaded per sub-batch
    proc = StackProcessor(6, 2, 512, 640, num_iter=100000, search_channels=(0, 1), mode="P", n_streams=3)
    outs.append(proc(stack.cpu().pin_memory(), 0.9, 100.0, seed=5))
    for other in outs[1:]:
        for a, b in zip(outs[0]["beads"], other["beads"]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(outs[0]["sums"].cpu().numpy(), other["sums"].cpu().numpy())
    assert sum(len(b) for b in outs[0]["beads"]) > 50


def test_full_size_assays_match_c_oracle(mg):
    """BASELINE's plane size (4096 x 4096, 5e6 RANSAC iterations): the whole chain of two assays
    against the oracle's C restatement -- corrected image, bead tables, ROI windows, masks and sums
    all bit-exact."""
    from magnify_amd.stack import StackProcessor, synthetic_stack
    from oracle import cport

    T, C, S = 2, 2, 4096
    stack, _ = synthetic_stack(T, C, S, S, seed=4100)
    flat_np = vignette((S, S))
    proc = StackProcessor(T, C, S, S, num_iter=5_000_000, search_channels=(0,), mode="P")
    # a first call (other seeds) goes through the checked chain; the call that is compared then runs with ONE host
    # round trip, on the sweeps / rounds / capacities of the first (hotpath.CircleFinder.find)
    proc(stack, torch.from_numpy(flat_np).cuda(), 100.0, seed=5)
    assert not proc.finder.stats["optimistic"]
    out = proc(stack, torch.from_numpy(flat_np).cuda(), 100.0, seed=9)
    assert proc.finder.stats["optimistic"] and proc.finder.calls == {"optimistic": 1, "repaired": 0, "checked": 1, "followed_again": 0}
    host = stack.cpu().numpy()
    off = out["offsets"]
    for t in range(T):
        img = cport.flatfield_correct(host[t][:, None, None, None], flat_np, 100.0)[:, 0, 0, 0]
        np.testing.assert_array_equal(proc.image[t].cpu().numpy(), img)
        want = cport.bead_assay(img, proc.min_r, proc.max_r, proc.L, num_iter=5_000_000,
                                seed=(9 + 1000003 * t) & 0xFFFFFFFFFFFFFFFF)
        np.testing.assert_array_equal(out["beads"][t], want["beads"])
        assert len(want["beads"]) > 1500
        lo, hi = off[t], off[t + 1]
        np.testing.assert_array_equal(out["roi"][lo:hi, :, 0].cpu().numpy(), want["roi"])
        np.testing.assert_array_equal(out["fg"][lo:hi].cpu().numpy().astype(bool), want["fg"])
        np.testing.assert_array_equal(out["bg"][lo:hi].cpu().numpy().astype(bool), want["bg"])
        sums = out["sums"][lo:hi, :, 0].cpu().numpy()
        np.testing.assert_array_equal(sums[..., 0], want["fg_sum"])
        np.testing.assert_array_equal(sums[..., 1], want["bg_sum"])
        np.testing.assert_array_equal(out["counts"][lo:hi].cpu().numpy(), np.stack([want["fg_count"], want["bg_count"]], 1))


def test_mode_r_matches_oracle_small(mg):
    """Single-assay semantics (find.py:477, 543-550): detection on time 0, maxima over the stack."""
    from magnify_amd.stack import StackProcessor
    from synth import noisy_bead_image

    planes = np.stack([np.stack([noisy_bead_image(50 + 3 * t + c, (256, 256), 6, r_lo=6, r_hi=10)[0] for c in range(2)])
                       for t in range(3)])  # (T, C, H, W)
    proc = StackProcessor(3, 2, 256, 256, num_iter=20000, min_bead_diameter=10, max_bead_diameter=24,
                          search_channels=(0,), mode="R")
    out = proc(torch.from_numpy(planes).cuda(), 0.9, 90.0, seed=3)
    want_img = rp.flatfield_correct(planes, 0.9, 90.0)  # maxima over the whole stack
    np.testing.assert_array_equal(proc.image.cpu().numpy(), want_img)
    assert len(out["beads"]) == 1 and out["roi"].shape[1:3] == (2, 3)
    # geometry comes from time 0 only and the windows of every time point are gathered
    b = out["beads"][0]
    assert len(b) >= 4
    from oracle import ref_numeric as rn

    for i in range(len(b)):
        top, bottom, left, right = rn.bounding_box(int(b[i, 1]), int(b[i, 0]), proc.L, 256, 256)
        want = want_img[:, :, top:bottom, left:right].transpose(1, 0, 2, 3)  # (C, T, L, L)
        np.testing.assert_array_equal(out["roi"][i].cpu().numpy(), want)


def test_stitched_chip_pipeline(mg):
    """C3 shape at reduced scale: a chip canvas cut into 4 x 4 overlapping tiles, stitched, grid-fit."""
    canvas = draw_chip((8, 8), 20, row_dist=250, col_dist=250)  # 2250 x 2250
    ty = tx = 600
    overlap = 50
    step = ty - overlap
    clip = overlap // 2
    tiles = np.zeros((4, 4, ty, tx), dtype=np.uint16)
    for r in range(4):
        for c in range(4):
            tiles[r, c] = canvas[r * step : r * step + ty, c * step : c * step + tx]
    data = mg.DataArray(tiles, ("row", "col", "y", "x"))
    xp = mg.microfluidic_chip(data=data, shape=(8, 8), overlap=overlap, row_dist=250, col_dist=250,
                              min_button_diameter=16, max_button_diameter=32, num_iter=200000)
    assert xp.image.shape == (4 * step, 4 * step)
    np.testing.assert_array_equal(xp.image.values, canvas[clip : clip + 4 * step, clip : clip + 4 * step])
    xp = xp.unstack().transpose("mark_row", "mark_col", ...)
    for i in range(8):
        for j in range(8):
            assert abs(xp.x[i, j].values.item() - (250 * (j + 1) - clip)) < 8
            assert abs(xp.y[i, j].values.item() - (250 * (i + 1) - clip)) < 8
    radii = np.sqrt(xp.fg.sum(["roi_x", "roi_y"]).to_numpy() / np.pi)
    assert 0.85 * 10 < radii.min() and radii.max() < 1.15 * 10
    assert xp.mg.cache(["roi"]) is xp


def test_argument_errors(mg):
    from magnify_amd import hotpath as hp

    img = torch.zeros((1, 1, 1, 32, 32), dtype=torch.uint16, device="cuda")
    with pytest.raises(ValueError):
        hp.roi_gather_reduce(img, [np.array([[5, 5, 3]])], 64, None)  # window larger than the image
    with pytest.raises(ValueError):
        hp.CircleFinder(1, 32, 32, 10, 5, 100)
    with pytest.raises(ValueError):
        mg.beads(mg.DataArray(np.zeros((64, 64), np.uint16), ("y", "x")), min_bead_diameter=30, max_bead_diameter=20,
                 overlap=0, num_iter=10)
    with pytest.raises(TypeError):
        mg.beads(mg.DataArray(np.zeros((64, 64), np.complex64), ("y", "x")), overlap=0, num_iter=10)


def test_full_size_chip_stages_match_c_oracle(mg, monkeypatch):
    """BASELINE's chip size (C3: 28 x 28 buttons on a 7376 x 7376 stitched image, 5e6 RANSAC iterations):
    ButtonFinder's grid fit and per-chamber refinement against the oracle's restatement of find_centers /
    find_rois, with the oracle's C port doing the circle search (the NumPy one takes minutes at this size)."""
    from magnify_amd.find import ButtonFinder
    from oracle import cport

    def c_find_circles(img_u8, low_q, high_q, grid, num_iter, min_r, max_r, min_roundness, min_dist, seed=0, **_):
        return cport.find_circles(img_u8, low_q, high_q, grid, num_iter, min_r, max_r, min_roundness, min_dist, seed=seed)

    monkeypatch.setattr(rp, "find_circles", c_find_circles)
    n, pitch = 28, 250
    canvas = draw_chip((n, n), 20, row_dist=pitch, col_dist=pitch)
    img = np.zeros((7376, 7376), dtype=np.uint16)
    img[: min(7376, canvas.shape[0]), : min(7376, canvas.shape[1])] = canvas[:7376, :7376]
    img[img > 0] += 123
    tag = np.full((n, n), "default", dtype="<U200")
    tag[3, 4] = tag[20, 11] = ""
    bf = ButtonFinder(row_dist=pitch, col_dist=pitch, min_button_diameter=8, max_button_diameter=30, chamber_diameter=60,
                      top_chamber=None, left_chamber=None, low_edge_quantile=0.1, high_edge_quantile=0.9,
                      num_iter=5_000_000, min_roundness=0.2, cluster_penalty=50, roi_length=None, progress_bar=False,
                      search_timestep=0, search_channel=None, interactive=False)
    d_img = torch.from_numpy(img).cuda()
    gx, gy = bf.find_centers(d_img[None], tag, [4242])
    ox, oy = rp.find_centers(img[None], tag, pitch, pitch, 4, 15, 30, 0.1, 0.9, 5_000_000, 0.2, 50, seed=4242)
    np.testing.assert_allclose(gx, ox, rtol=0, atol=1e-9)
    np.testing.assert_allclose(gy, oy, rtol=0, atol=1e-9)
    x, y, radius = bf.refine(d_img[None], gx, gy, tag, [0], 777)
    _, fg_o, _, x_o, y_o = rp.find_rois(img[None], ox, oy, tag, [0], 4, 15, 30, 72, 0.1, 5_000_000, 0.2, seed=777)
    np.testing.assert_allclose(x, x_o, atol=1e-9)
    np.testing.assert_allclose(y, y_o, atol=1e-9)
    found = tag != ""
    # the oracle's fg mask is cv.circle's disk of the refined radius: same radii <=> same pixel counts
    from oracle import ref_opencv as rcv

    area = {r: int(rcv.filled_circle_mask((72, 72), (36, 36), r).sum()) for r in range(4, 16)}
    np.testing.assert_array_equal(fg_o.sum(axis=(-1, -2)), np.vectorize(area.get)(radius))
    assert (np.abs(x[found] - pitch * (np.arange(n)[None, :] + 1).repeat(n, 0)[found]) <= 2).all()
    assert radius[3, 4] == 15 and (radius[found] >= 8).all() and (radius[found] <= 12).all()


def test_public_api_c2_matches_c_oracle(mg):
    """BASELINE's C2 (4 ch x 4096^2, reference-default 5e6 iterations) through the drop-in call mg.beads:
    bead positions, ROI pixels and fg / bg masks equal the oracle's C restatement."""
    from magnify_amd.stack import synthetic_stack
    from oracle import cport

    stack, _ = synthetic_stack(1, 4, 4096, 4096, seed=2000)
    planes = stack[0]
    mg.seed(2100)
    xp = mg.beads(data=mg.DataArray(planes, ("channel", "y", "x")), overlap=0, num_iter=5_000_000, search_channel=0)
    host = planes.cpu().numpy()
    img = cport.flatfield_correct(host[:, None, None, None], 1.0, 0.0)[:, 0, 0, 0]
    first = (2100 + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF  # the first seed magnify_amd.utils.next_seed hands out
    want = cport.bead_assay(img, 5, 25, 100, num_iter=5_000_000, seed=first)
    m = xp.roi.sizes["mark"]
    assert m == len(want["beads"]) > 1500
    np.testing.assert_array_equal(np.asarray(xp.y.values).reshape(m, -1)[:, 0], want["beads"][:, 0])
    np.testing.assert_array_equal(np.asarray(xp.x.values).reshape(m, -1)[:, 0], want["beads"][:, 1])
    np.testing.assert_array_equal(np.asarray(xp.roi.values).reshape(want["roi"].shape), want["roi"])
    np.testing.assert_array_equal(np.asarray(xp.fg.values).reshape(want["fg"].shape), want["fg"])
    np.testing.assert_array_equal(np.asarray(xp.bg.values).reshape(want["bg"].shape), want["bg"])


def test_streamed_series_equals_whole_stack(mg, tmp_path):
    """Config C5's path at small scale: TIFF files -> reader.iter_time_chunks -> stack.process_stream
    (reader thread ahead of the GPU) gives, chunk by chunk, exactly what one StackProcessor call gives on
    the whole stack -- whatever the chunk size."""
    from PIL import Image

    from magnify_amd import reader
    from magnify_amd.stack import StackProcessor, process_stream, synthetic_stack

    T, C, S = 7, 2, 320
    stack, _ = synthetic_stack(T, C, S, S, seed=515)
    host = stack.cpu().numpy()
    for t in range(T):
        for c in range(C):
            Image.fromarray(host[t, c]).save(tmp_path / f"x_ch{c}_202401{t + 10}.tif")
    kw = dict(num_iter=60000, search_channels=(0,))
    whole = StackProcessor(T, C, S, S, mode="P", **kw)(stack, 0.9, 90.0, seed=21)
    whole_sums = whole["sums"].cpu().numpy()  # a view of a pooled buffer: the next processor call reuses it
    pattern = str(tmp_path / "x_(channel)_(time|%Y%m%d).tif")
    for chunk in (3, 7, 2):
        got_beads, got_sums, firsts = [], [], []
        for out in process_stream(reader.iter_time_chunks(pattern, chunk, pinned=True), 0.9, 90.0, seed=21, **kw):
            assert out["channel"] == ["ch0", "ch1"]
            got_beads += out["beads"]
            got_sums.append(out["sums"].cpu().numpy())
            firsts.append(out["first_timepoint"])
        assert firsts == list(range(0, T, chunk)) and len(got_beads) == T
        for a, b in zip(got_beads, whole["beads"]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(np.concatenate(got_sums), whole_sums)
    assert sum(len(b) for b in whole["beads"]) > 30


def test_streamed_tiled_series_equals_eager_pipeline(mg, tmp_path):
    """Config C5's path with TILES: (row) / (col) / (channel) / (time) files -> reader.iter_time_chunks (one page
    at a time, chunked by time) -> stack.process_stream(overlap=...): the flat-field pass crops and joins the
    tiles on the device, so the stitched assay never exists on the host.  Per timepoint the result equals the
    eager path (Reader -> flatfield_correct -> Stitcher -> StackProcessor on the stitched stack)."""
    from PIL import Image

    from magnify_amd import reader
    from magnify_amd.stack import StackProcessor, process_stream, stitched_shape, synthetic_stack

    T, C, R, Cc, ty, ov = 5, 2, 4, 4, 96, 10
    step = ty - ov
    side = (R - 1) * step + ty
    canvas, _ = synthetic_stack(T, C, side, side, seed=616, beads_per_mpx=400.0)
    canvas = canvas.cpu().numpy()
    tiles = np.empty((T, C, R, Cc, ty, ty), dtype=np.uint16)
    for r in range(R):
        for c in range(Cc):
            tiles[:, :, r, c] = canvas[:, :, r * step : r * step + ty, c * step : c * step + ty]
            for t in range(T):
                for ch in range(C):
                    Image.fromarray(tiles[t, ch, r, c]).save(tmp_path / f"a_ch{ch}_202402{t + 10}_r{r}_c{c}.tif")
    pattern = str(tmp_path / "a_(channel)_(time|%Y%m%d)_r(row)_c(col).tif")
    kw = dict(num_iter=60000, search_channels=(0,))
    # eager: the whole assay through the registered reader, then the oracle's stitch of the corrected tiles
    xp = list(reader.Reader()(pattern))[0]
    eager_tiles = np.asarray(xp.tile.values).transpose(1, 0, 2, 3, 4, 5)  # (T, C, R, Cc, ty, tx)
    np.testing.assert_array_equal(eager_tiles, tiles)
    h, w = stitched_shape(R, Cc, ty, ty, ov)
    whole = StackProcessor(T, C, h, w, mode="P", tile_grid=(R, Cc, ty, ty), overlap=ov, **kw)
    want = whole(torch.from_numpy(tiles).cuda(), 0.9, 90.0, seed=31)
    want_sums = want["sums"].cpu().numpy()
    for t in range(T):  # the device's stitched image == oracle flat-field (per assay) + stitch
        ref = rp.stitch(rp.flatfield_correct(tiles[t][:, None], 0.9, 90.0), ov)[:, 0]
        np.testing.assert_array_equal(whole.image[t].cpu().numpy(), ref)
    assert sum(len(b) for b in want["beads"]) > 30
    for chunk in (2, 5):
        beads, sums = [], []
        for out in process_stream(reader.iter_time_chunks(pattern, chunk, pinned=True), 0.9, 90.0, seed=31, overlap=ov, **kw):
            assert out["channel"] == ["ch0", "ch1"]
            beads += out["beads"]
            sums.append(out["sums"].cpu().numpy())
        assert len(beads) == T
        for a, b in zip(beads, want["beads"]):
            np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(np.concatenate(sums), want_sums)


def test_stream_sinks_persist_every_assay(mg, tmp_path):
    """the results of a streamed run persist.  process_stream(sink=...) copies every chunk's ROI
    pixels, masks and reductions to the host beside the next chunk's kernels (two sets of pooled device buffers in
    turn) and turns every timepoint into a Dataset with the reference's schema -- kept (HostSink) or written with
    mg.save and read back with mg.load (SaveSink) -- equal, assay for assay, to one StackProcessor call on the stack."""
    from magnify_amd.stack import StackProcessor, process_stream, synthetic_stack

    T, C, S, L = 7, 2, 320, 40
    stack, _ = synthetic_stack(T, C, S, S, seed=717, beads_per_mpx=350.0)
    kw = dict(num_iter=60000, search_channels=(0,), roi_length=L, min_bead_diameter=10, max_bead_diameter=44)
    whole = StackProcessor(T, C, S, S, mode="P", **kw)(stack, 0.9, 90.0, seed=5)
    want = {k: whole[k].cpu().numpy().copy() for k in ("roi", "fg", "bg", "sums", "counts")}
    off, beads = whole["offsets"], whole["beads"]
    assert off[-1] > 40

    def chunks(n):
        for lo in range(0, T, n):
            yield list(range(100 + lo, 100 + min(lo + n, T))), ["dna", "cy5"], stack[lo: lo + n].cpu()

    def check(ds, t):
        lo, hi = off[t], off[t + 1]
        np.testing.assert_array_equal(ds["roi"].values, want["roi"][lo:hi])
        np.testing.assert_array_equal(ds.coords["fg"].values[:, 0], want["fg"][lo:hi].astype(bool))
        np.testing.assert_array_equal(ds.coords["bg"].values[:, 0], want["bg"][lo:hi].astype(bool))
        np.testing.assert_array_equal(ds["fg_sum"].values, want["sums"][lo:hi, ..., 0])
        np.testing.assert_array_equal(ds["bg_count"].values, want["counts"][lo:hi, 1])
        np.testing.assert_array_equal(ds.coords["x"].values[:, 0], beads[t][:, 1])
        np.testing.assert_array_equal(ds.coords["y"].values[:, 0], beads[t][:, 0])
        assert ds["roi"].dims == ("mark", "channel", "time", "roi_y", "roi_x") and list(ds.coords["channel"].values) == ["dna", "cy5"]
        assert int(ds.coords["time"].values[0]) == 100 + t

    host = mg.HostSink()
    n_out = sum(1 for _ in process_stream(chunks(2), 0.9, 90.0, seed=5, want_roi=True, sink=host, **kw))  # 4 chunks: 2, 2, 2, 1
    assert n_out == 4 and sorted(host.assays) == list(range(T))
    for t in range(T):
        check(host.assays[t], t)
    assert host.assays[0]["roi"].chunksizes["mark"][0] == min(len(beads[0]), -(-10**6 // (L * L * C)))
    saver = mg.SaveSink(str(tmp_path / "series_t{index:04d}.nc"))
    for _ in process_stream(chunks(3), 0.9, 90.0, seed=5, want_roi=True, sink=saver, **kw):
        pass
    assert sorted(saver.files) == list(range(T))
    for t in range(T):
        check(mg.load(saver.files[t]), t)
    light = mg.HostSink(want_roi=False, want_masks=False)  # reductions only: nothing but tables crosses the bus
    for _ in process_stream(chunks(7), 0.9, 90.0, seed=5, sink=light, **kw):
        pass
    assert "roi" not in light.assays[3].data_vars
    np.testing.assert_array_equal(light.assays[3]["fg_sum"].values, want["sums"][off[3]:off[4], ..., 0])


def test_save_and_spill_a_gpu_resident_result(mg, tmp_path):
    """mg.save of a result whose roi / fg / bg still live in HBM (file.py:6-17 on the output of mg.beads), and
    Dataset.mg.cache(spill=...) -- the reference's zarr spill (accessor.py:18-35) as a move to page-locked host memory
    or to memory-mapped files: same values, other backing."""
    from synth import noisy_bead_image

    img = np.stack([noisy_bead_image(900 + c, (512, 640), 25, r_lo=6, r_hi=12)[0] for c in range(2)])
    xp = mg.beads(data=mg.DataArray(data=img, dims=("channel", "y", "x")), min_bead_diameter=10, max_bead_diameter=30,
                  overlap=0, num_iter=100000, search_channel=0)
    m = xp.roi.sizes["mark"]
    assert m >= 15 and torch.is_tensor(xp.roi.raw) and xp.roi.raw.is_cuda  # device-resident
    assert xp.roi.chunksizes["mark"][0] == min(m, -(-10**6 // (60 * 60 * 2)))
    ref = {k: np.asarray(xp[k].values).copy() for k in ("roi", "fg", "bg", "x", "y", "valid")}
    mg.save(tmp_path / "beads.nc", xp)
    back = mg.load(tmp_path / "beads.nc")
    for k, v in ref.items():
        np.testing.assert_array_equal(np.asarray(back[k].values), v, err_msg=k)
        assert back[k].dims == xp[k].dims
    mg.save(tmp_path / "parts.nc", xp, shard_bytes=200000)  # parts cut from the device-resident arrays
    np.testing.assert_array_equal(mg.load(tmp_path / "parts.nc")["roi"].values, ref["roi"])
    xp.mg.cache(["roi", "fg"], spill="host")
    assert isinstance(xp.roi.raw, np.ndarray) and isinstance(xp.fg.raw, np.ndarray) and torch.is_tensor(xp.bg.raw)
    xp.mg.cache("bg", spill="disk")
    assert isinstance(xp.variables["bg"].raw, np.memmap)  # backed by a file in the process's spill directory
    for k, v in ref.items():
        np.testing.assert_array_equal(np.asarray(xp[k].values), v, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["P", "R"])
def test_placement_trial_changes_nothing_but_the_blocks(mg, monkeypatch, mode):
    """The first call of a large mode-P processor times its passes into several image blocks / ROI output sets and keeps
    the fastest (StackProcessor._placement_tries): the results of that call and of the next are those of a processor
    that never tried (MG_PLACEMENT_TRIES=0), and the blocks not kept are gone."""
    import torch

    from magnify_amd import hotpath as hp
    from magnify_amd.stack import StackProcessor, synthetic_stack

    T, C, S = 17, 4, 4096  # 2.3 GB of image: above the trial's threshold
    stack, _ = synthetic_stack(T, C, S, S, seed=77)
    kw = dict(num_iter=200_000, search_channels=(0,), mode=mode)  # (R: the host-table route of the ROI pass, one assay)

    def run(tries):
        monkeypatch.setenv("MG_PLACEMENT_TRIES", tries)
        hp.release_pool()
        proc = StackProcessor(T, C, S, S, **kw)
        outs = []
        for seed in (3, 4):
            o = proc(stack, 1.0, 100.0, seed=seed)
            outs.append({k: (o[k].clone() if torch.is_tensor(o[k]) else o[k]) for k in ("roi", "fg", "bg", "sums", "counts", "beads")})
        return proc, outs

    plain, want = run("0")
    assert plain.placement is None
    del plain
    tried, got = run("3,3")
    assert len(tried.placement["flatfield_ms"]) == 3 and np.asarray(tried.placement["roi_ms"]).shape == (3, 3)
    assert tried._trial_blocks is None
    assert tried.pool_tag == "#place%d" % tried.placement["roi_set"]
    assert not [k for k in hp._POOL if k[0] in ("roi", "roi#place%d" % ((tried.placement["roi_set"] + 1) % 3))]
    for a, b in zip(want, got):
        m = sum(len(x) for x in a["beads"])
        assert m == sum(len(x) for x in b["beads"]) and all(np.array_equal(x, y) for x, y in zip(a["beads"], b["beads"]))
        for key in ("roi", "fg", "bg", "sums", "counts"):
            assert torch.equal(a[key][:m], b[key][:m]), key
