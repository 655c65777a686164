"""The oracle's C restatement (oracle/c/ref_port.c) against the golden vectors produced by the
reference's own code and against the NumPy oracle.  Integer / index work is bit-exact; the two
libm-dependent spots (float64 atan2 of perimeter offsets, float32 arctan2 of the gradient) are
compared at 1 ulp of float32 on the scores."""
import numpy as np
import pytest

from oracle import cport
from oracle import ref_numeric as rn
from oracle import ref_opencv as cv
from oracle import ref_pipeline as rp
from synth import draw_beads, noisy_bead_image


def angles64(u8):
    """float32 gradient angle rounded from a float64 arctan2: what the C port (and the HIP kernel)
    computes; NumPy's own float32 arctan2 is a SIMD kernel that is up to 2 ulp off."""
    dx, dy = cv.scharr(cv.gaussian_blur5(u8))
    return np.arctan2(dy.astype(np.float64), dx.astype(np.float64)).astype(np.float32)


def ulp_diff_f32(a, b):
    a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return int(np.abs(a - b).max()) if a.size else 0


# ---- against the reference's golden vectors ------------------------------------------------------

def test_circle_tables_golden(golden):
    g = golden("circle_tables")
    for r in g["radii"]:
        r = int(r)
        np.testing.assert_array_equal(cport.circle_points(r), g[f"p8_{r}"])
        np.testing.assert_array_equal(cport.circle_points(r, True), g[f"p4_{r}"])
        np.testing.assert_array_equal(cport.filled_circle_points(r), g[f"fill_{r}"])
    np.testing.assert_array_equal(cport.circle_points(1), g["p8_1"])
    with pytest.raises(ValueError):
        cport.filled_circle_points(1)


def test_bounding_box_golden(golden):
    for x, y, length, w, h, *exp in golden("bounding_box")["cases"]:
        assert cport.bounding_box(int(x), int(y), int(length), int(w), int(h)) == tuple(int(v) for v in exp)


def test_to_uint8_golden(golden):
    g = golden("to_uint8")
    for k in ("a16", "a16n", "const", "small"):
        if g[k].dtype == np.uint16:
            np.testing.assert_array_equal(cport.to_uint8(g[k]), g[k + "_out"])


def test_grid_and_candidates_golden(golden):
    g = golden("grid_candidates")
    coords, starts, counts = cport.grid_array(g["edges"], 20)
    np.testing.assert_array_equal(coords, g["gcoords"])
    np.testing.assert_array_equal(starts, g["starts"])
    np.testing.assert_array_equal(counts, g["counts"])
    cand = cport.candidate_circles_from_picks(g["edges"], 20, g["i0"], g["j1"], g["j2"])
    np.testing.assert_array_equal(cand.view(np.uint32), g["candidates"].view(np.uint32))  # NaN / inf included


def test_mean_grad_golden(golden):
    g = golden("mean_grad")
    for rad in (8, 9, 10, 12):
        s = cport.mean_grad(g["angles"], g["edges"], g[f"centers_{rad}"], rad)
        assert ulp_diff_f32(s, g[f"sums_{rad}"]) <= 1  # libm atan2 of the offsets vs NumPy's


def test_filter_neighbors_golden(golden):
    g = golden("filter_neighbors")
    for case in range(4):
        v = cport.filter_neighbors(g[f"circles_{case}"], int(g[f"min_dist_{case}"]))
        np.testing.assert_array_equal(v, g[f"valid_{case}"])
    np.testing.assert_array_equal(cport.filter_neighbors(np.array([[30, 30, 10], [32, 31, 10], [60, 60, 9]]), 8),
                                  g["anchor"])


def test_circle_labels_golden(golden):
    g = golden("circle_labels")
    h, w = g["shape"]
    np.testing.assert_array_equal(cport.circle_labels(g["beads"], int(h), int(w)), g["labels"])


# ---- against the NumPy oracle -----------------------------------------------------------------------

@pytest.mark.parametrize("shape,seed", [((300, 340), 3), ((257, 129), 4), ((64, 48), 5), ((7, 9), 6), ((1, 40), 7),
                                        ((33, 1), 8)])
def test_edge_stage_matches_numpy(shape, seed):
    rng = np.random.default_rng(seed)
    if min(shape) >= 64:
        img, _ = noisy_bead_image(seed, shape, 6, r_lo=6, r_hi=12)
    else:
        img = rng.integers(0, 4000, shape).astype(np.uint16)
    u8 = rn.to_uint8(img)
    np.testing.assert_array_equal(cport.to_uint8(img), u8)
    np.testing.assert_array_equal(cport.gaussian_blur5(u8), cv.gaussian_blur5(u8))
    blur, dx, dy, edges, lohi = rp.edge_stage(u8, 0.1, 0.9)
    cb, cdx, cdy, ce, clohi = cport.edge_stage(u8, 0.1, 0.9)
    np.testing.assert_array_equal(cb, blur)
    np.testing.assert_array_equal(cdx, dx.astype(np.int16))
    np.testing.assert_array_equal(cdy, dy.astype(np.int16))
    assert clohi == lohi
    np.testing.assert_array_equal(ce, edges)


def test_quantile_matches_numpy():
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 10, 1001, 65537, 300000):
        v = np.sqrt(rng.integers(0, 33_000_000, n).astype(np.float32))
        for q in (0.0, 0.1, 0.5, 0.9, 0.99731, 1.0, 1 - np.pi * 4 / 72**2):
            assert cport.quantile_f32(v, q) == np.quantile(v, q), (n, q)
    v = np.zeros(5000, np.float32)  # the reference tests' degenerate case: both quantiles 0
    assert cport.quantile_f32(v, 0.9) == 0


def test_noiseless_image_like_reference_tests():
    img = draw_beads((200, 220), [[60, 60], [120, 150], [150, 60]], 20, 1000)
    u8 = rn.to_uint8(img)
    oc, osc = rp.find_circles(u8, 0.1, 0.9, 20, 20000, 5, 25, 0.3, 5, seed=11, grad_angles=angles64(u8))
    cc, csc = cport.find_circles(u8, 0.1, 0.9, 20, 20000, 5, 25, 0.3, 5, seed=11)
    np.testing.assert_array_equal(cc, oc)
    assert ulp_diff_f32(csc, osc) <= 1
    assert len(cc) == 3


@pytest.mark.parametrize("seed", [7, 8, 9])
def test_find_circles_matches_numpy(seed):
    img, _ = noisy_bead_image(seed, (300, 340), 10, r_lo=6, r_hi=12)
    u8 = rn.to_uint8(img)
    oc, osc = rp.find_circles(u8, 0.1, 0.9, 20, 30000, 5, 13, 0.3, 5, seed=seed, grad_angles=angles64(u8))
    cc, csc = cport.find_circles(u8, 0.1, 0.9, 20, 30000, 5, 13, 0.3, 5, seed=seed)
    np.testing.assert_array_equal(cc, oc)
    assert ulp_diff_f32(csc, osc) <= 1
    # no suppression: the full scored list, in canonical order
    oc, osc = rp.find_circles(u8, 0.1, 0.9, 20, 30000, 5, 13, 0.3, 0, seed=seed, grad_angles=angles64(u8))
    cc, csc = cport.find_circles(u8, 0.1, 0.9, 20, 30000, 5, 13, 0.3, 0, seed=seed)
    assert len(cc) == len(oc)
    same = (cc == oc).all(axis=1)
    assert same.mean() > 0.999  # a 1-ulp score difference may swap two near-tied neighbours
    assert ulp_diff_f32(np.sort(csc), np.sort(osc)) <= 1


def test_empty_and_flat_images():
    z = np.zeros((50, 60), np.uint8)
    c, s = cport.find_circles(z, 0.1, 0.9, 20, 1000, 5, 13, 0.3, 5)
    assert c.shape == (0, 3) and s.shape == (0,)
    out = cport.bead_assay(np.zeros((2, 50, 60), np.uint16), 5, 13, 26, num_iter=1000)
    assert len(out["beads"]) == 0 and out["roi"].shape == (0, 2, 26, 26)


def test_flatfield_matches_numpy():
    rng = np.random.default_rng(2)
    tiles = rng.integers(0, 65535, (3, 2, 1, 1, 40, 50)).astype(np.uint16)
    flat = (0.7 + 0.3 * rng.random((40, 50))).astype(np.float32)
    np.testing.assert_array_equal(cport.flatfield_correct(tiles, flat, 100.0), rp.flatfield_correct(tiles, flat, 100.0))
    np.testing.assert_array_equal(cport.flatfield_correct(tiles, 0.9, 3.5), rp.flatfield_correct(tiles, 0.9, 3.5))
    np.testing.assert_array_equal(cport.flatfield_correct(tiles, 1.0, 0.0), rp.flatfield_correct(tiles, 1.0, 0.0))


def test_bead_assay_matches_numpy():
    planes = np.stack([noisy_bead_image(20 + c, (300, 340), 10, r_lo=6, r_hi=12)[0] for c in range(3)])
    image = planes[:, None]  # (C, T=1, H, W)
    want = rp.find_beads(image, 10, 26, num_iter=30000, search_channels=[0, 2], seed=5)
    red = rp.roi_reduce(want["roi"], want["fg"], want["bg"], medians=False)
    got = cport.bead_assay(planes, 5, 13, 52, num_iter=30000, search_channels=(0, 2), seed=5)
    np.testing.assert_array_equal(got["beads"], want["beads"])
    np.testing.assert_array_equal(got["roi"], want["roi"][:, :, 0])
    np.testing.assert_array_equal(got["fg"], want["fg"][:, 0])
    np.testing.assert_array_equal(got["bg"], want["bg"][:, 0])
    np.testing.assert_array_equal(got["fg_sum"], red["fg_sum"][:, :, 0])
    np.testing.assert_array_equal(got["bg_sum"], red["bg_sum"][:, :, 0])
    np.testing.assert_array_equal(got["fg_count"], red["fg_count"][:, 0])
    np.testing.assert_array_equal(got["bg_count"], red["bg_count"][:, 0])
    assert len(got["beads"]) >= 8


def test_run_stack_is_the_per_assay_loop():
    stack = np.stack([np.stack([noisy_bead_image(40 + 4 * t + c, (200, 240), 5, r_lo=6, r_hi=12)[0] for c in range(2)])
                      for t in range(3)])
    flat = (1 - 0.3 * np.linspace(0, 1, 200 * 240).reshape(200, 240) ** 2).astype(np.float32)
    seeds = [100, 200, 300]
    total, counts, sums = cport.run_stack(stack, flat, 100.0, 5, 13, 52, seeds, num_iter=20000, n_threads=2)
    for t in range(3):
        img = cport.flatfield_correct(stack[t][:, None, None, None], flat, 100.0)[:, 0, 0, 0]
        one = cport.bead_assay(img, 5, 13, 52, num_iter=20000, seed=seeds[t], want_roi=False)
        assert counts[t] == len(one["beads"]) and sums[t] == one["fg_sum"].sum()
    assert total == counts.sum() and total > 0
