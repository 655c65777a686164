"""Generate the golden fixtures in this directory from the reference's own code.

Run in the build container only (``python tests/golden/make_golden.py``); needs
``/root/reference``.  Executes the reference's ``utils.py`` / ``find.py`` in place
through ``_ref_harness`` and stores *inputs and expected outputs* as small
``.npz`` files.  No reference source is stored.  The unseeded
``np.random.choice`` draws inside ``candidate_circles`` (utils.py:311-320) are
replayed from explicit pick lists so that the arithmetic is reproducible.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness  # noqa: E402

utils, find = _ref_harness.load()


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)} bytes")


def circles_tables():
    radii = np.array([2, 3, 4, 5, 7, 8, 10, 12, 15, 16, 20, 25, 30, 33], dtype=np.int64)
    out = {"radii": radii}
    for r in radii:
        out[f"p8_{r}"] = utils.circle_points(int(r))
        out[f"p4_{r}"] = utils.circle_points(int(r), four_connected=True)
        out[f"fill_{r}"] = utils.filled_circle_points(int(r))
    out["p8_1"] = utils.circle_points(1)
    out["p4_1"] = utils.circle_points(1, four_connected=True)
    save("circle_tables", **out)


def boxes():
    rng = np.random.default_rng(11)
    cases = []
    for _ in range(200):
        w, h = int(rng.integers(100, 5000)), int(rng.integers(100, 5000))
        length = int(rng.integers(1, 100))
        x, y = int(rng.integers(-50, w + 50)), int(rng.integers(-50, h + 50))
        cases.append([x, y, length, w, h, *utils.bounding_box(x, y, length, w, h)])
    cases += [[0, 0, 100, 4096, 4096, *utils.bounding_box(0, 0, 100, 4096, 4096)],
              [4095, 4095, 100, 4096, 4096, *utils.bounding_box(4095, 4095, 100, 4096, 4096)],
              [2000, 2000, 101, 4096, 4096, *utils.bounding_box(2000, 2000, 101, 4096, 4096)]]
    save("bounding_box", cases=np.array(cases, dtype=np.int64))


def uint8_cases():
    rng = np.random.default_rng(12)
    a16 = rng.integers(0, 65536, size=(64, 80), dtype=np.uint16)
    a16n = rng.integers(90, 4000, size=(50, 70), dtype=np.uint16)
    af32 = (rng.normal(1000, 300, size=(40, 60))).astype(np.float32)
    af64 = rng.random((30, 30))
    const = np.full((8, 8), 77, dtype=np.uint16)
    small = np.array([0, 1, 2, 3, 65535], dtype=np.uint16)
    save("to_uint8",
         a16=a16, a16_out=utils.to_uint8(a16), a16n=a16n, a16n_out=utils.to_uint8(a16n),
         af32=af32, af32_out=utils.to_uint8(af32), af64=af64, af64_out=utils.to_uint8(af64),
         const=const, const_out=utils.to_uint8(const), small=small, small_out=utils.to_uint8(small),
         empty_out=utils.to_uint8(np.zeros((0, 5), dtype=np.uint16)))


def ring_edges(shape, beads, rng, noise=0.002):
    """A 0/1 edge map made of perimeter rings plus sparse noise (test input only)."""
    e = (rng.random(shape) < noise).astype(np.uint8)
    for (r0, c0, rad) in beads:
        pts = utils.circle_points(int(rad)) + np.array([r0, c0])
        ok = (pts[:, 0] >= 0) & (pts[:, 0] < shape[0]) & (pts[:, 1] >= 0) & (pts[:, 1] < shape[1])
        e[pts[ok, 0], pts[ok, 1]] = 1
    return e


def grid_and_candidates():
    rng = np.random.default_rng(13)
    shape = (97, 113)
    beads = [(30, 30, 10), (60, 80, 12), (20, 90, 8), (85, 40, 9), (5, 5, 10)]
    edges = ring_edges(shape, beads, rng)
    gcoords, starts, counts = utils.grid_array(edges, 20)
    # Explicit picks, replayed through the reference's np.random.choice calls.
    coords = np.column_stack(np.where(edges))
    k = 3000
    i0 = rng.integers(0, len(coords), size=k)
    cells = coords[i0] // 20
    cnt = counts[cells[:, 0], cells[:, 1]]
    j1 = (rng.random(k) * cnt).astype(np.int64)
    j2 = (rng.random(k) * cnt).astype(np.int64)
    # Force some degenerate draws: p1 == p0, p2 == p1, collinear-ish.
    j2[:50] = j1[:50]
    stream = np.stack([i0, j1, j2], axis=1).reshape(-1).tolist()
    it = iter(stream)
    real_choice = np.random.choice
    np.random.choice = lambda n, *a, **kw: int(next(it))
    try:
        with np.errstate(all="ignore"):
            cand = utils.candidate_circles(edges, 20, k)
    finally:
        np.random.choice = real_choice
    save("grid_candidates", edges=edges, gcoords=gcoords, starts=starts, counts=counts,
         i0=i0, j1=j1, j2=j2, candidates=cand,
         empty=utils.candidate_circles(np.zeros((40, 40), dtype=np.uint8), 20, 10))


def scoring():
    rng = np.random.default_rng(14)
    h, w, max_r = 120, 140, 12
    pad = 2 * max_r
    beads = [(40, 40, 10), (70, 100, 12), (100, 30, 8), (2, 70, 9)]
    edges = ring_edges((h, w), beads, rng, noise=0.01)
    # Radial-ish angles near beads, random elsewhere.
    yy, xx = np.mgrid[0:h, 0:w]
    angles = rng.uniform(-np.pi, np.pi, size=(h, w)).astype(np.float32)
    for (r0, c0, rad) in beads:
        near = np.abs(np.hypot(yy - r0, xx - c0) - rad) < 1.5
        angles[near] = np.arctan2(yy - r0, xx - c0).astype(np.float32)[near]
    pa = np.pad(angles, pad)
    pe = np.pad(edges, pad)
    out = {"edges": edges, "angles": angles, "pad": np.int64(pad)}
    for rad in (8, 9, 10, 12):
        n = 60
        centers = np.stack([rng.integers(-rad, h + rad, n), rng.integers(-rad, w + rad, n)], axis=1)
        for k, (r0, c0, rr) in enumerate(beads):
            centers[k] = (r0 + rng.integers(-1, 2), c0 + rng.integers(-1, 2))
        centers = centers.astype(np.int32)
        per = utils.circle_points(rad)
        s = utils.mean_grad(pa, pe, centers + pad, per)
        out[f"centers_{rad}"] = centers
        out[f"sums_{rad}"] = s
    save("mean_grad", **out)


def nms():
    rng = np.random.default_rng(15)
    out = {}
    for case, (n, span, min_dist) in enumerate([(400, 120, 5), (300, 80, 8), (200, 300, 30), (50, 40, 4)]):
        c = np.stack([rng.integers(-6, span, n), rng.integers(-6, span, n), rng.integers(5, 26, n)], axis=1)
        c = c.astype(np.int32)
        out[f"circles_{case}"] = c
        out[f"min_dist_{case}"] = np.int64(min_dist)
        out[f"valid_{case}"] = utils.filter_neighbors(c, min_dist)
    out["anchor"] = utils.filter_neighbors(np.array([[30, 30, 10], [32, 31, 10], [60, 60, 9]]), 8)
    save("filter_neighbors", **out)


def labels():
    rng = np.random.default_rng(16)
    h, w = 150, 170
    n = 40
    beads = np.stack([rng.integers(-5, h + 5, n), rng.integers(-5, w + 5, n), rng.integers(2, 14, n)], axis=1)
    beads[:3] = [[50, 50, 10], [55, 58, 10], [52, 54, 6]]  # overlapping trio
    lab = utils.circle_labels(beads.astype(int), h, w)
    save("circle_labels", beads=beads.astype(np.int64), shape=np.array([h, w]), labels=lab)


def clusters():
    rng = np.random.default_rng(17)
    out = {}
    # A 6 x 5 grid with jitter, a few missing and a few spurious points.
    n_rows, n_cols, rd, cd = 6, 5, 100.0, 120.5
    gy, gx = np.mgrid[0:n_rows, 0:n_cols]
    y = (150 + gy * rd + 0.02 * gx * cd + rng.normal(0, 2, gy.shape)).ravel()
    x = (130 + gx * cd - 0.02 * gy * rd + rng.normal(0, 2, gx.shape)).ravel()
    keep = rng.random(len(x)) > 0.1
    y = np.concatenate([y[keep], rng.uniform(0, 900, 3)])
    x = np.concatenate([x[keep], rng.uniform(0, 900, 3)])
    ideal_r = np.full(n_rows, n_cols)
    ideal_c = np.full(n_cols, n_rows)
    rl = find.cluster_1d(y, total_length=900, num_clusters=n_rows, cluster_length=rd,
                         ideal_num_points=ideal_r, penalty=50)
    cl = find.cluster_1d(x, total_length=900, num_clusters=n_cols, cluster_length=cd,
                         ideal_num_points=ideal_c, penalty=50)
    out.update(x=x, y=y, ideal_r=ideal_r, ideal_c=ideal_c, rd=rd, cd=cd, row_labels=rl, col_labels=cl)
    out["row_labels_fixed"] = find.label_clusters(y, offset=120, num_clusters=n_rows, cluster_length=60,
                                                  cluster_gap=rd - 60)
    inside = (rl >= 0) & (cl >= 0)
    xs, ys, rls, cls = x[inside], y[inside], rl[inside], cl[inside]
    s, b = find.regress_clusters(xs, ys, labels=rls, num_clusters=n_rows, ideal_num_points=ideal_r)
    out.update(row_slope=np.float64(s), row_intercepts=np.asarray(b, dtype=np.float64))
    s, b = find.regress_clusters(ys, xs, labels=cls, num_clusters=n_cols, ideal_num_points=ideal_c)
    out.update(col_slope=np.float64(s), col_intercepts=np.asarray(b, dtype=np.float64))
    # Single cluster variants.
    x1 = np.array([10.0, 50.0, 90.0, 130.0])
    y1 = np.array([100.0, 101.0, 103.5, 104.0])
    s, b = find.regress_clusters(x1, y1, labels=np.zeros(4, int), num_clusters=1, ideal_num_points=np.array([4]))
    out.update(x1=x1, y1=y1, single_slope=np.float64(s), single_intercept=np.float64(b))
    # Ragged clusters: sizes 2 .. 9, one EMPTY cluster (2), one with a single point (5), unsorted labels, a few points
    # outside every cluster (-1); lines with a common slope plus noise.
    sizes = [4, 9, 0, 2, 7, 1, 5]
    rx, ry, rlab = [], [], []
    for i, k in enumerate(sizes):
        px = rng.uniform(50, 950, k)
        rx.append(px)
        ry.append(120.0 * i + 0.013 * px + rng.normal(0, 1.5, k))
        rlab.append(np.full(k, i))
    rx, ry, rlab = np.concatenate(rx + [rng.uniform(0, 900, 3)]), np.concatenate(ry + [rng.uniform(0, 900, 3)]), \
        np.concatenate(rlab + [np.full(3, -1)]).astype(int)
    perm = rng.permutation(len(rx))
    rx, ry, rlab = rx[perm], ry[perm], rlab[perm]
    ideal_rag = np.array([5, 9, 4, 0, 7, 3, 5])
    s, b = find.regress_clusters(rx, ry, labels=rlab, num_clusters=len(sizes), ideal_num_points=ideal_rag)
    out.update(rag_x=rx, rag_y=ry, rag_labels=rlab, rag_ideal=ideal_rag, rag_slope=np.float64(s),
               rag_intercepts=np.asarray(b, dtype=np.float64))
    save("clusters", **out)


if __name__ == "__main__":
    circles_tables()
    boxes()
    uint8_cases()
    grid_and_candidates()
    scoring()
    nms()
    labels()
    clusters()
