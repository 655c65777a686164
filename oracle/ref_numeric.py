"""Oracle: the reference's numeric helpers (``src/magnify/utils.py``), restated.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Every function cites the
reference lines it follows; all are pinned bit-for-bit by ``tests/golden``.
The restatements are vectorised NumPy where the reference runs numba loops;
floating-point work keeps the reference's operation order and dtypes so that the
results are bit-identical, not merely close.
"""
from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------------------
# scalar helpers
# --------------------------------------------------------------------------------------


def to_uint8(arr: np.ndarray) -> np.ndarray:
    """Global min/max rescale to [0,255] with truncating cast (utils.py:20-27)."""
    arr = np.asarray(arr)
    if arr.size == 0:
        return arr.astype(np.uint8)
    a = arr.astype(np.float64)
    a = a - a.min()
    top = a.max()
    if top > 0:
        a = 255 * a / top
    return a.astype(np.uint8)


def ceildiv(a: int, b: int) -> int:
    """utils.py:55-57."""
    return -(a // -b)


def bounding_box(x: int, y: int, box_length: int, image_width: int, image_height: int):
    """Clamp an L x L window centred on (x, y) into the image (utils.py:60-80).

    Returns (top, bottom, left, right).  Note the argument order (x, y, L, W, H).
    """
    lo_half = box_length // 2
    hi_half = ceildiv(box_length, 2)

    def clamp(c, size):
        a, b = c - lo_half, c + hi_half
        if a < 0:
            a, b = 0, b - a
        if b > size:
            a, b = a - (b - size), size
        return a, b

    top, bottom = clamp(y, image_height)
    left, right = clamp(x, image_width)
    return top, bottom, left, right


# --------------------------------------------------------------------------------------
# digital circles
# --------------------------------------------------------------------------------------


def circle_points(r: int, four_connected: bool = False) -> np.ndarray:
    """Perimeter offsets (row, col) of the midpoint circle, in the reference's
    emission order (utils.py:433-465).  Order matters for float summation."""
    pts = [(0, -r), (-r, 0), (0, r), (r, 0)]
    x, y = 1, -r
    while x < -y:
        pts += [(x, y), (y, x), (-x, y), (-y, x), (x, -y), (y, -x), (-x, -y), (-y, -x)]
        if x * x + y * y - r * r <= 0:
            x += 1
        else:
            y += 1
            if not four_connected:
                x += 1
    if y == -x:
        pts += [(x, y), (-x, -y), (-x, y), (x, -y)]
    return np.asarray(pts, dtype=np.int32).reshape(-1, 2)


def filled_circle_points(r: int) -> np.ndarray:
    """Perimeter followed by the row-wise interior fill (utils.py:398-430).

    r < 2 is out of bounds in the reference (IndexError / ValueError in plain
    Python, silent under numba); the oracle raises ValueError."""
    if r < 2:
        raise ValueError("filled_circle_points is undefined for r < 2 in the reference")
    per = circle_points(r)
    size = 2 * r + 1
    mask = np.zeros((size, size), dtype=bool)
    mask[per[:, 0] + r, per[:, 1] + r] = True
    interior = []
    for i in range(size):
        row = mask[i]
        j = 0
        while not row[j]:
            j += 1
        while row[j]:
            j += 1
        if j <= r:
            while not row[j]:
                interior.append((i - r, j - r))
                j += 1
    if interior:
        return np.concatenate([per, np.asarray(interior, dtype=np.int32)])
    return per


def circle_labels(circles: np.ndarray, num_rows: int, num_cols: int) -> np.ndarray:
    """Ownership map: -1 nobody, i exactly bead i, -2 contested (utils.py:380-395)."""
    labels = np.full((num_rows, num_cols), -1, dtype=np.int32)
    for i in range(len(circles)):
        pts = filled_circle_points(int(circles[i, 2])) + np.asarray(circles[i, :2], dtype=np.int64)
        ok = (pts[:, 0] >= 0) & (pts[:, 0] < num_rows) & (pts[:, 1] >= 0) & (pts[:, 1] < num_cols)
        rr, cc = pts[ok, 0], pts[ok, 1]
        # Points of one disk are distinct, so "already owned" can only mean another bead.
        taken = labels[rr, cc] != -1
        labels[rr[taken], cc[taken]] = -2
        labels[rr[~taken], cc[~taken]] = i
    return labels


# --------------------------------------------------------------------------------------
# edge grid (CSR by 20x20 cell) and RANSAC circle candidates
# --------------------------------------------------------------------------------------


def grid_array(arr: np.ndarray, grid_length: int):
    """Per-cell edge counts, CSR starts and coordinates, cell-major and row-major
    inside a cell (utils.py:347-377)."""
    n_rows = math.ceil(arr.shape[0] / grid_length)
    n_cols = math.ceil(arr.shape[1] / grid_length)
    r, c = np.nonzero(arr)
    cell = (r // grid_length) * n_cols + (c // grid_length)
    order = np.lexsort((c, r, cell))  # primary: cell, then row, then col
    coords = np.stack([r[order], c[order]], axis=1).astype(np.int32)
    # The reference counts with arr[...].sum(): identical to the nonzero count for a 0/1 map.
    weights = arr[r, c].astype(np.int64)
    counts = np.bincount(cell, weights=weights, minlength=n_rows * n_cols).astype(np.int64)
    counts = counts.reshape(n_rows, n_cols)
    nz_counts = np.bincount(cell, minlength=n_rows * n_cols).astype(np.int64)
    starts = (np.cumsum(nz_counts) - nz_counts).reshape(n_rows, n_cols)
    return coords, starts, counts


def circumcircles(p0: np.ndarray, p1: np.ndarray, p2: np.ndarray) -> np.ndarray:
    """Circle through three pixel coordinates, in the reference's arithmetic
    (utils.py:319-342).  p* are (K, 2) integer (row, col) arrays in image
    coordinates; returns (K, 3) float32 (row, col, r).

    dtype walk-through of the reference (NumPy promotion, as executed under the
    golden harness): p1-p0 is int64; 0.5f * int64 -> float64; int64 + float32
    eps -> float64; every intermediate is float64; each *store* into the float32
    ``circles`` array rounds once, and circles[i,1] is re-read as float32 before
    it is used for circles[i,0]; the radius is float32 arithmetic throughout;
    the final re-centring is float32 + int64 -> float64 -> float32 store.
    """
    p0 = np.asarray(p0, dtype=np.int64)
    q1 = np.asarray(p1, dtype=np.int64) - p0
    q2 = np.asarray(p2, dtype=np.int64) - p0
    eps = np.float64(np.float32(1e-20))
    with np.errstate(all="ignore"):
        mid1 = 0.5 * q1.astype(np.float64)
        mid2 = 0.5 * q2.astype(np.float64)
        m1 = (-q1[:, 1]).astype(np.float64) / (q1[:, 0].astype(np.float64) + eps)
        m2 = (-q2[:, 1]).astype(np.float64) / (q2[:, 0].astype(np.float64) + eps)
        b1 = mid1[:, 0] - m1 * mid1[:, 1]
        b2 = mid2[:, 0] - m2 * mid2[:, 1]
        c_col = ((b1 - b2) / (m2 - m1 + eps)).astype(np.float32)
        c_row = (m1 * c_col.astype(np.float64) + b1).astype(np.float32)
        rad = np.sqrt(c_row * c_row + c_col * c_col)  # float32 throughout
        out = np.empty((len(p0), 3), dtype=np.float32)
        out[:, 0] = (c_row.astype(np.float64) + p0[:, 0].astype(np.float64)).astype(np.float32)
        out[:, 1] = (c_col.astype(np.float64) + p0[:, 1].astype(np.float64)).astype(np.float32)
        out[:, 2] = rad
    return out


def candidate_circles_from_picks(edges: np.ndarray, grid_length: int, i0, j1, j2) -> np.ndarray:
    """``candidate_circles`` (utils.py:295-344) with the three random draws made
    explicit: per iteration ``i0`` indexes the row-major edge list
    (``np.where(edges)``), ``j1``/``j2`` index p0's grid cell list.  The
    reference draws them with an unseeded ``np.random.choice`` (not reproducible)."""
    rows, cols = np.nonzero(edges)
    if len(rows) == 0:
        return np.empty((0, 3), dtype=np.float32)
    coords = np.stack([rows, cols], axis=1)
    gcoords, starts, counts = grid_array(edges, grid_length)
    p0 = coords[np.asarray(i0, dtype=np.int64)]
    cell_r, cell_c = p0[:, 0] // grid_length, p0[:, 1] // grid_length
    base = starts[cell_r, cell_c]
    p1 = gcoords[base + np.asarray(j1, dtype=np.int64)]
    p2 = gcoords[base + np.asarray(j2, dtype=np.int64)]
    return circumcircles(p0, p1, p2)


# --------------------------------------------------------------------------------------
# scoring and greedy non-maximum suppression
# --------------------------------------------------------------------------------------


def mean_grad(grad_angles: np.ndarray, edges: np.ndarray, centers: np.ndarray, perimeter: np.ndarray):
    """Sum over perimeter edge pixels of the radial-alignment score (utils.py:225-251).

    Accumulation is float64 and sequential in perimeter order per circle (the
    reference's inner prange is serial), stored to float32.  ``centers`` must be
    inside the padded arrays."""
    centers = np.asarray(centers, dtype=np.int64)
    expected = np.arctan2(perimeter[:, 0], perimeter[:, 1])  # int32 -> float64
    acc = np.zeros(len(centers), dtype=np.float64)
    for j in range(len(perimeter)):
        rr = centers[:, 0] + int(perimeter[j, 0])
        cc = centers[:, 1] + int(perimeter[j, 1])
        on = edges[rr, cc] > 0
        d = np.abs(grad_angles[rr, cc].astype(np.float64) - expected[j])
        d = np.where(d > np.pi, d - np.pi, d)
        term = 4 * np.abs(d - np.pi / 2) / np.pi - 1
        acc = np.where(on, acc + term, acc)
    return acc.astype(np.float32)


def filter_neighbors(circles: np.ndarray, min_dist: int) -> np.ndarray:
    """Greedy score-ordered suppression on a claim grid (utils.py:254-292).

    A circle is dropped when any pixel of the 4-connected *ring* of radius
    ``min_dist`` about its centre is already claimed, otherwise it claims the
    ring.  Negative grid indices wrap, as they do under numba."""
    n = len(circles)
    if n == 0:
        return np.ones(0, dtype=bool)
    circles = np.asarray(circles, dtype=np.int64)
    ring = circle_points(min_dist, four_connected=True).astype(np.int64)
    pad = 2 * min_dist + 1
    n_rows = int(circles[:, 0].max()) + 2 * pad
    n_cols = int(circles[:, 1].max()) + 2 * pad
    claimed = np.zeros((n_rows, n_cols), dtype=bool)
    keep = np.ones(n, dtype=bool)
    for i in range(n):
        rr = (ring[:, 0] + circles[i, 0] + pad) % n_rows
        cc = (ring[:, 1] + circles[i, 1] + pad) % n_cols
        if claimed[rr, cc].any():
            keep[i] = False
        else:
            claimed[rr, cc] = True
    return keep


# --------------------------------------------------------------------------------------
# filter_circles (steps 4-6 of find_circles)
# --------------------------------------------------------------------------------------


CANON_TILE = 64  # side of the centre tiles of the build's canonical circle order


def canonical_key(circles: np.ndarray, max_radius: int):
    """Sort keys (minor first, for np.lexsort) of the build's canonical circle order: centres are
    binned into 64 x 64 tiles of the padded centre grid (row + max_radius, col + max_radius); order is
    (tile_row, tile_col, r, row, col).  This is the order in which the GPU emits unique circles."""
    c = np.asarray(circles, dtype=np.int64)
    tr = (c[:, 0] + max_radius) // CANON_TILE
    tc = (c[:, 1] + max_radius) // CANON_TILE
    return (c[:, 1], c[:, 0], c[:, 2], tc, tr)


def canonical_order(circles: np.ndarray, scores: np.ndarray, max_radius: int) -> np.ndarray:
    """The build's canonical total order for score ties: score descending, then the tile-major
    circle order of ``canonical_key``.  The reference uses an unstable argsort (utils.py:195), so
    any tie order is a legal outcome of it; duplicates of one integer circle always tie, and
    dropping them does not change the survivors of ``filter_neighbors`` (a duplicate of a kept
    circle hits its own ring, a duplicate of a dropped circle is dropped for the same reason)."""
    return np.lexsort(canonical_key(circles, max_radius) + (-scores.astype(np.float64),))


def filter_circles(all_circles, edges, dx, dy, min_radius, max_radius, min_roundness, min_dist,
                   dedup=True, grad_angles=None):
    """Steps 4-6 of ``find_circles`` (utils.py:149-199).

    ``grad_angles`` may be supplied by a test to replace ``np.arctan2(dy, dx)``
    (float32), which is platform-dependent in its last bit."""
    h, w = edges.shape
    rad = all_circles[:, 2]
    with np.errstate(invalid="ignore"):
        c = all_circles[(rad >= min_radius) & (rad <= max_radius)]
        c = np.round(c).astype(np.int32)
    c = c[(c[:, 0] + c[:, 2] >= 0) & (c[:, 1] + c[:, 2] >= 0) & (c[:, 0] - c[:, 2] < h) & (c[:, 1] - c[:, 2] < w)]
    if dedup:
        c = np.unique(c, axis=0)
    if grad_angles is None:
        grad_angles = np.arctan2(dy, dx)
    pad = 2 * max_radius
    ang = np.pad(grad_angles, pad)
    pedges = np.pad(edges, pad)
    order = np.lexsort((c[:, 1], c[:, 0], c[:, 2]))  # by radius (then row, col: canonical)
    c = c[order]
    scores = np.empty(len(c), dtype=np.float32)
    start = 0
    for radius in range(min_radius, max_radius + 1):
        per = circle_points(radius)
        end = int(np.searchsorted(c[:, 2], radius + 1))
        s = mean_grad(ang, pedges, c[start:end, :2] + pad, per)
        scores[start:end] = s / len(per)
        start = end
    good = scores >= min_roundness
    c, scores = c[good], scores[good]
    perm = canonical_order(c, scores, max_radius)
    c, scores = c[perm], scores[perm]
    if min_dist > 0:
        keep = filter_neighbors(c, min_dist)
        c, scores = c[keep], scores[keep]
    return c, scores


# --------------------------------------------------------------------------------------
# the build's explicit RNG stream (the reference is unseeded)
# --------------------------------------------------------------------------------------

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z ^ (z >> np.uint64(30))
    z = z * _M1
    z = z ^ (z >> np.uint64(27))
    z = z * _M2
    return z ^ (z >> np.uint64(31))


def draw_uniform32(seed: int, it: np.ndarray, k: int) -> np.ndarray:
    """k-th (k = 0, 1, 2) 32-bit uniform of iteration ``it`` for plane seed ``seed``: splitmix64's finaliser applied
    to seed + (3*it + c) * golden -- k = 0: the top half at counter c = 1; k = 1 and k = 2: the top and the bottom half
    at counter c = 2 (one finaliser for the two picks inside p0's cell; counter c = 3 is not used)."""
    with np.errstate(over="ignore"):
        ctr = np.asarray(it, dtype=np.uint64) * np.uint64(3) + np.uint64(1 if k == 0 else 2)
        z = _mix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + ctr * _GOLDEN)
        return (z & np.uint64(0xFFFFFFFF)) if k == 2 else (z >> np.uint64(32))


def stratum(it: np.ndarray, n_edges: int, num_iter: int):
    """Iteration ``it`` owns the slice [a, b) of the cell-major edge list, a = floor(it * E / K),
    b = floor((it + 1) * E / K) (width at least 1)."""
    it = np.asarray(it, dtype=np.uint64)
    a = (it * np.uint64(n_edges)) // np.uint64(num_iter)
    b = ((it + np.uint64(1)) * np.uint64(n_edges)) // np.uint64(num_iter)
    return a, np.maximum(b - a, np.uint64(1))


def draw_picks(seed: int, num_iter: int, edges: np.ndarray, grid_length: int):
    """Map the RNG stream to the three picks of every iteration.

    Definition shared with the HIP kernel.  p0 is a jittered stratified draw over the cell-major
    edge list: iteration i picks p0 = list[a + (u0 * width >> 32)] inside its own slice
    ``stratum(i)`` -- every edge pixel is equally likely, as with the reference's iid
    ``np.random.choice`` (utils.py:311), and any such set of draws is a possible outcome of the
    unseeded reference, but consecutive iterations are spatially coherent.  p1, p2 = that cell's
    list[u * count >> 32] (utils.py:318-321).  Returns picks in the *reference's* indexing (i0 into
    the row-major list, j1/j2 into the cell list) so that they can be replayed through
    ``candidate_circles_from_picks``."""
    gcoords, starts, counts = grid_array(edges, grid_length)
    n_edges = len(gcoords)
    it = np.arange(num_iter, dtype=np.uint64)
    if n_edges == 0:
        z = np.zeros(0, dtype=np.int64)
        return z, z, z
    a, width = stratum(it, n_edges, num_iter)
    u0 = a + ((draw_uniform32(seed, it, 0) * width) >> np.uint64(32))
    p0 = gcoords[u0.astype(np.int64)].astype(np.int64)
    cnt = counts[p0[:, 0] // grid_length, p0[:, 1] // grid_length].astype(np.uint64)
    j1 = ((draw_uniform32(seed, it, 1) * cnt) >> np.uint64(32)).astype(np.int64)
    j2 = ((draw_uniform32(seed, it, 2) * cnt) >> np.uint64(32)).astype(np.int64)
    # Row-major rank of p0 (what the reference's coords[...] index would have been).
    w = edges.shape[1]
    rows, cols = np.nonzero(edges)
    flat_sorted = rows.astype(np.int64) * w + cols
    i0 = np.searchsorted(flat_sorted, p0[:, 0] * w + p0[:, 1])
    return i0.astype(np.int64), j1, j2
