"""BASELINE.json's configs C1 and C3 at full size through the drop-in API, checked against the oracle
(C2 is tests/test_gpu_fullsize.py::test_public_api_c2_matches_c_oracle, C4 is bench.py's workload and
test_full_size_assays_match_c_oracle, C5's path is the streamed / tiled tests).  The work is
tests/config_table.py's, which also times it for DESIGN.md's table."""
import argparse

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _args():
    return argparse.Namespace(num_iter=5_000_000, repeat=1, no_cpu=False)


def test_c1_single_plane_through_mg_beads():
    """C1: one 2048 x 2048 single-channel plane (seed 1000, 503 drawn beads), reference-default 5e6
    iterations, through mg.beads: bead positions, ROI pixels and fg masks equal the C oracle's."""
    import config_table

    rec = config_table.bead_config("C1", 1, 2048, 1000, _args())
    assert rec["same_beads_roi_fg_as_gpu"] is True
    assert rec["markers"] >= 0.8 * rec["drawn_beads"] and rec["shape"] == [1, 1, 2048, 2048]


def test_c3_tile_stack_through_mg_microfluidic_chip():
    """C3: the 8 x 8 stack of 1024^2 tiles (overlap 102 -> 7376^2), 28 x 28 buttons, through
    mg.microfluidic_chip: the stitched image equals the oracle's stitch, all 784 positions equal the
    oracle's find_centers / find_rois (its C port as the circle search) and sit on the drawn grid."""
    import config_table

    rec = config_table.chip_config(_args())
    assert rec["same_image_as_oracle_stitch"] is True
    assert rec["same_xy_as_gpu"] is True
    assert rec["markers"] == 784 and rec["stitched"] == [7376, 7376]
    assert rec["max_centre_error_px"] <= 1.0


def test_c5_tiled_series_streamed_at_scale():
    """C5: the C4 generator as a TILED acquisition -- 64 timepoints x 4 channels x 4 x 4 tiles of 1126^2
    with overlap 102 (-> 4096^2 stitched), 10.4 GB of tile bytes -- streamed chunk by chunk (8 timepoints) from
    page-locked host memory through stack.process_stream: a chunk is generated, handed over and forgotten; the tiles
    are cropped / joined / corrected on the device, so neither the series nor a stitched assay ever exists on the
    host.  Two timepoints, from different chunks, are checked in full against the C oracle (its flat-field over the
    assay's tiles, the NumPy stitch, its bead search with the reference-default 5e6 iterations and its ROI
    reductions: same beads, same sums, same counts); every timepoint by properties of the result (beads inside the
    image or reaching into it, apart by at least the suppression distance, brighter than their background, foreground areas that fit
    their radii; marker counts steady along the series)."""
    from oracle import cport
    from oracle import ref_pipeline as rp
    from magnify_amd import HostSink
    from magnify_amd.stack import process_stream, stitched_shape, synthetic_stack

    T, C, R, ty, ov, chunk = 64, 4, 4, 1126, 102, 8
    h, w = stitched_shape(R, R, ty, ty, ov)
    assert (h, w) == (4096, 4096)
    step = ty - 2 * (ov // 2) - ov % 2
    side = (R - 1) * step + ty
    yy, xx = np.mgrid[0:ty, 0:ty]
    flat = (1 - 0.15 * (((yy - (ty - 1) / 2) / (ty / 2)) ** 2 + ((xx - (ty - 1) / 2) / (ty / 2)) ** 2)).astype(np.float32)
    sample, kept, drawn = (3, 42), {}, {}

    def series():
        for t0 in range(0, T, chunk):
            canvas, truth = synthetic_stack(chunk, C, side, side, seed=5000 + t0)
            tiles = torch.empty((chunk, C, R, R, ty, ty), dtype=torch.uint16, device="cuda")
            for r in range(R):
                for c in range(R):
                    tiles[:, :, r, c] = canvas[:, :, r * step:r * step + ty, c * step:c * step + ty]
            block = tiles.cpu().pin_memory()
            for t in sample:
                if t0 <= t < t0 + chunk:
                    kept[t] = block[t - t0].numpy().copy()
            drawn[t0] = len(truth)
            del canvas, tiles
            yield list(range(t0, t0 + chunk)), [f"ch{c}" for c in range(C)], block

    sink = HostSink(want_roi=False, want_masks=False)
    firsts = [out["first_timepoint"] for out in process_stream(series(), flat, 100.0, seed=7, overlap=ov, sink=sink, prefetch=1,
                                                               num_iter=5_000_000, search_channels=(0,))]
    assert firsts == list(range(0, T, chunk)) and sorted(sink.assays) == list(range(T))
    counts = np.array([sink.assays[t]["radius"].shape[0] for t in range(T)])
    assert counts.min() > 0.85 * min(drawn.values()) and counts.max() < 1.1 * max(drawn.values())
    for t in range(T):
        ds = sink.assays[t]
        x, y, r = ds.coords["x"].values[:, 0], ds.coords["y"].values[:, 0], ds["radius"].values
        # (a circle may be centred off the image as long as it reaches into it, utils.py:161-166)
        assert (x + r >= 0).all() and (x - r < w).all() and (y + r >= 0).all() and (y - r < h).all() and (r >= 5).all() and (r <= 25).all()
        order = np.lexsort((x, y))
        pts = np.column_stack([y, x])[order]
        near = pts[1:] - pts[:-1]
        assert not ((np.abs(near[:, 0]) < 1) & (np.abs(near[:, 1]) < 5)).any()  # no two kept circles on top of each other
        fg_n, bg_n = ds["fg_count"].values.astype(np.float64), ds["bg_count"].values.astype(np.float64)
        assert (fg_n <= np.pi * (r + 1.5) ** 2).all() and (fg_n + bg_n <= 100 * 100).all()
        fg_mean = ds["fg_sum"].values[:, 0, 0] / np.maximum(fg_n, 1)
        bg_mean = ds["bg_sum"].values[:, 0, 0] / np.maximum(bg_n, 1)
        assert (fg_mean > bg_mean)[fg_n > 0].mean() > 0.9  # (the last few per cent of a table are circles found in noise)
    for t in sample:  # the full comparison
        tiles = kept[t]                                                    # (C, R, R, ty, tx)
        corrected = cport.flatfield_correct(tiles, flat, 100.0)           # one assay: maxima over all its tiles
        image = rp.stitch(corrected[:, None], ov)[:, 0]                   # (C, 4096, 4096)
        want = cport.bead_assay(image, 5, 25, 100, num_iter=5_000_000, seed=(7 + 1000003 * t) & 0xFFFFFFFFFFFFFFFF, want_roi=False)
        ds = sink.assays[t]
        got = np.column_stack([ds.coords["y"].values[:, 0], ds.coords["x"].values[:, 0], ds["radius"].values]).astype(np.int32)
        np.testing.assert_array_equal(got, want["beads"], err_msg=f"timepoint {t}")
        np.testing.assert_array_equal(ds["fg_sum"].values[:, :, 0], want["fg_sum"])
        np.testing.assert_array_equal(ds["bg_sum"].values[:, :, 0], want["bg_sum"])
        np.testing.assert_array_equal(ds["fg_count"].values, want["fg_count"])
        np.testing.assert_array_equal(ds["bg_count"].values, want["bg_count"])
