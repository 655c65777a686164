"""The oracle against the golden vectors produced by the reference's own code
(tests/golden/make_golden.py).  Bit-exact everywhere."""
import numpy as np
import pytest

from oracle import ref_numeric as rn
from oracle import ref_pipeline as rp


def test_circle_tables(golden):
    g = golden("circle_tables")
    for r in g["radii"]:
        r = int(r)
        np.testing.assert_array_equal(rn.circle_points(r), g[f"p8_{r}"])
        np.testing.assert_array_equal(rn.circle_points(r, True), g[f"p4_{r}"])
        np.testing.assert_array_equal(rn.filled_circle_points(r), g[f"fill_{r}"])
    np.testing.assert_array_equal(rn.circle_points(1), g["p8_1"])
    np.testing.assert_array_equal(rn.circle_points(1, True), g["p4_1"])
    # SURVEY 8c known-answer anchors: 8-conn / 4-conn / filled sizes.
    for r, (a, b, c) in {2: (12, 16, 21), 3: (16, 24, 37), 5: (32, 40, 93), 8: (48, 64, 217), 10: (60, 80, 341),
                         12: (68, 96, 473), 16: (92, 128, 837), 25: (144, 200, 2021)}.items():
        assert (len(rn.circle_points(r)), len(rn.circle_points(r, True)), len(rn.filled_circle_points(r))) == (a, b, c)
    with pytest.raises(ValueError):
        rn.filled_circle_points(1)


def test_disk_points_are_distinct():
    # circle_labels' coverage-count formulation relies on this.
    for r in range(2, 70):
        pts = rn.filled_circle_points(r)
        assert len(np.unique(pts, axis=0)) == len(pts)


def test_bounding_box(golden):
    for x, y, length, w, h, *exp in golden("bounding_box")["cases"]:
        assert rn.bounding_box(int(x), int(y), int(length), int(w), int(h)) == tuple(int(v) for v in exp)


def test_to_uint8(golden):
    g = golden("to_uint8")
    for k in ("a16", "a16n", "af32", "af64", "const", "small"):
        out = rn.to_uint8(g[k])
        assert out.dtype == np.uint8
        np.testing.assert_array_equal(out, g[k + "_out"])
    assert rn.to_uint8(np.zeros((0, 5), dtype=np.uint16)).shape == g["empty_out"].shape


def test_grid_array(golden):
    g = golden("grid_candidates")
    coords, starts, counts = rn.grid_array(g["edges"], 20)
    np.testing.assert_array_equal(coords, g["gcoords"])
    assert coords.dtype == g["gcoords"].dtype
    np.testing.assert_array_equal(starts, g["starts"])
    np.testing.assert_array_equal(counts, g["counts"])


def test_candidate_circles(golden):
    g = golden("grid_candidates")
    cand = rn.candidate_circles_from_picks(g["edges"], 20, g["i0"], g["j1"], g["j2"])
    assert cand.dtype == np.float32 and cand.shape == g["candidates"].shape
    # Bit-exact, NaN/inf included (degenerate picks).
    np.testing.assert_array_equal(cand.view(np.uint32), g["candidates"].view(np.uint32))
    assert rn.candidate_circles_from_picks(np.zeros((40, 40), np.uint8), 20, [], [], []).shape == g["empty"].shape


def test_mean_grad(golden):
    g = golden("mean_grad")
    pad = int(g["pad"])
    pa, pe = np.pad(g["angles"], pad), np.pad(g["edges"], pad)
    for rad in (8, 9, 10, 12):
        s = rn.mean_grad(pa, pe, g[f"centers_{rad}"] + pad, rn.circle_points(rad))
        np.testing.assert_array_equal(s.view(np.uint32), g[f"sums_{rad}"].view(np.uint32))


def test_filter_neighbors(golden):
    g = golden("filter_neighbors")
    for case in range(4):
        v = rn.filter_neighbors(g[f"circles_{case}"], int(g[f"min_dist_{case}"]))
        np.testing.assert_array_equal(v, g[f"valid_{case}"])
    np.testing.assert_array_equal(rn.filter_neighbors(np.array([[30, 30, 10], [32, 31, 10], [60, 60, 9]]), 8),
                                  g["anchor"])
    assert rn.filter_neighbors(np.empty((0, 3), int), 5).shape == (0,)


def test_circle_labels(golden):
    g = golden("circle_labels")
    h, w = g["shape"]
    lab = rn.circle_labels(g["beads"], int(h), int(w))
    assert lab.dtype == np.int32
    np.testing.assert_array_equal(lab, g["labels"])
    assert (lab == -2).any() and (lab >= 0).any()


def test_clusters(golden):
    g = golden("clusters")
    rl = rp.cluster_1d(g["y"], 900, 6, float(g["rd"]), g["ideal_r"], 50)
    cl = rp.cluster_1d(g["x"], 900, 5, float(g["cd"]), g["ideal_c"], 50)
    np.testing.assert_array_equal(rl, g["row_labels"])
    np.testing.assert_array_equal(cl, g["col_labels"])
    np.testing.assert_array_equal(rp.label_clusters(g["y"], 120, 6, 60, float(g["rd"]) - 60), g["row_labels_fixed"])
    inside = (rl >= 0) & (cl >= 0)
    x, y = g["x"][inside], g["y"][inside]
    s, b = rp.regress_clusters(x, y, rl[inside], 6, g["ideal_r"])
    np.testing.assert_allclose(s, g["row_slope"], rtol=1e-12)
    np.testing.assert_allclose(b, g["row_intercepts"], rtol=1e-12)
    s, b = rp.regress_clusters(y, x, cl[inside], 5, g["ideal_c"])
    np.testing.assert_allclose(s, g["col_slope"], rtol=1e-12)
    np.testing.assert_allclose(b, g["col_intercepts"], rtol=1e-12)
    s, b = rp.regress_clusters(g["x1"], g["y1"], np.zeros(4, int), 1, np.array([4]))
    np.testing.assert_allclose([s, b], [g["single_slope"], g["single_intercept"]], rtol=1e-12)
    # ragged clusters with an empty one, a single-point one and points outside every cluster
    s, b = rp.regress_clusters(g["rag_x"], g["rag_y"], g["rag_labels"], 7, g["rag_ideal"])
    np.testing.assert_allclose(s, g["rag_slope"], rtol=1e-12)
    np.testing.assert_allclose(b, g["rag_intercepts"], rtol=1e-12)


def test_product_cluster_helpers_equal_the_reference_bit_for_bit(golden):
    """magnify_amd.find.regress_clusters / label_clusters (host-side product code, written independently of the
    oracle: linregress's own operations per cluster, one sort for the medians) against the values the REFERENCE's
    functions returned (tests/golden/make_golden.py): exactly equal, ragged / empty / single-point clusters included --
    the slopes feed a median and rounded chamber centres, an ulp can move a pixel."""
    from magnify_amd import find as product

    g = golden("clusters")
    rl, cl = g["row_labels"], g["col_labels"]
    inside = (rl >= 0) & (cl >= 0)
    x, y = g["x"][inside], g["y"][inside]
    s, b = product.regress_clusters(x, y, rl[inside], 6, g["ideal_r"])
    assert s == g["row_slope"]
    np.testing.assert_array_equal(b, g["row_intercepts"])
    s, b = product.regress_clusters(y, x, cl[inside], 5, g["ideal_c"])
    assert s == g["col_slope"]
    np.testing.assert_array_equal(b, g["col_intercepts"])
    s, b = product.regress_clusters(g["x1"], g["y1"], np.zeros(4, int), 1, np.array([4]))
    assert (s, b) == (g["single_slope"], g["single_intercept"])
    s, b = product.regress_clusters(g["rag_x"], g["rag_y"], g["rag_labels"], 7, g["rag_ideal"])
    assert s == g["rag_slope"]
    np.testing.assert_array_equal(b, g["rag_intercepts"])
    np.testing.assert_array_equal(product.label_clusters(g["y"], 120, 6, 60, float(g["rd"]) - 60), g["row_labels_fixed"])


def test_product_grouped_cluster_fits_equal_the_single_fits():
    """find._cluster_slopes (clusters of one size fitted together on stacked arrays) against find._line_fit (np.mean +
    np.cov per cluster, linregress's operations): the same bits for every cluster, over sizes 2 .. 70 and a chip-like
    mix of sizes; a cluster whose x are all equal is refused as linregress refuses it."""
    from magnify_amd import find as product

    rng = np.random.default_rng(77)
    sizes = np.concatenate([np.arange(2, 71), rng.integers(26, 29, size=56), [1, 0, 128, 257]])
    labels = np.repeat(np.arange(len(sizes)), sizes)
    perm = rng.permutation(len(labels))
    labels = labels[perm]
    x, y = rng.normal(3500, 2000, len(labels)), rng.normal(3500, 2000, len(labels))
    order0 = np.argsort(labels, kind="stable")
    bounds = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    got = product._cluster_slopes(x, y, order0, bounds, sizes.astype(np.int64), len(sizes))
    for i, n in enumerate(sizes):
        if n < 2:
            assert np.isnan(got[i])
        else:
            assert got[i] == product._line_fit(x[labels == i], y[labels == i])[0], (i, n)
    x[labels == 5] = 12.0
    with pytest.raises(ValueError):
        product._cluster_slopes(x, y, order0, bounds, sizes.astype(np.int64), len(sizes))
