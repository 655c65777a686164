"""Synthetic inputs for the tests and the bench (no reference code involved).

``draw_beads`` / ``draw_chip`` re-create the painters the reference's tests use
(tests/test_beads.py:9-36, tests/test_chip.py:9-34): constant-valued filled disks, with the
reference's ``filled_circle_points`` pixel set, on a zero background.
"""
import numpy as np

from oracle import ref_numeric as rn


def draw_beads(shape, positions, diameters=20, value=1000, dtype=np.uint16):
    positions = np.atleast_2d(np.asarray(positions))
    n = len(positions)
    diameters = np.full(n, diameters, dtype=np.int32) if np.isscalar(diameters) else np.asarray(diameters)
    values = np.full(n, value) if np.isscalar(value) else np.asarray(value)
    img = np.zeros(shape, dtype=dtype)
    for pos, d, v in zip(positions, diameters, values):
        pts = rn.filled_circle_points(int(d) // 2) + pos
        ok = (pts[:, 0] >= 0) & (pts[:, 0] < shape[0]) & (pts[:, 1] >= 0) & (pts[:, 1] < shape[1])
        img[pts[ok, 0], pts[ok, 1]] = v
    return img


def draw_chip(grid_shape, button_diameter=20, row_dist=100, col_dist=100, value=1000, blanks=(), offset=(0, 0)):
    rows, cols = grid_shape
    shape = ((rows + 1) * row_dist, (cols + 1) * col_dist)
    pos = [[(i + 1) * row_dist + offset[0], (j + 1) * col_dist + offset[1]]
           for i in range(rows) for j in range(cols) if (i, j) not in blanks]
    return draw_beads(shape, pos, button_diameter, value)


def random_bead_positions(rng, shape, n, r_max, border=None):
    """Rejection-sample n non-overlapping centres (min distance 2 r_max + 4)."""
    border = r_max + 2 if border is None else border
    pts = []
    tries = 0
    min_d2 = (2 * r_max + 4) ** 2
    while len(pts) < n and tries < 200 * n:
        tries += 1
        p = (int(rng.integers(border, shape[0] - border)), int(rng.integers(border, shape[1] - border)))
        if all((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2 >= min_d2 for q in pts):
            pts.append(p)
    return np.asarray(pts, dtype=np.int64).reshape(-1, 2)


def noisy_bead_image(seed, shape, n_beads, r_lo=8, r_hi=20, background=100, poisson=20.0, read_noise=3.0):
    """The BASELINE.md synthetic plane: uint16 background 100 + Poisson(20) + N(0,3) read noise,
    filled-disk beads of radius U{r_lo..r_hi} and value U{500..4000}."""
    rng = np.random.default_rng(seed)
    pos = random_bead_positions(rng, shape, n_beads, r_hi)
    radii = rng.integers(r_lo, r_hi + 1, size=len(pos))
    values = rng.integers(500, 4001, size=len(pos))
    img = background + rng.poisson(poisson, size=shape).astype(np.float64)
    beads = draw_beads(shape, pos, 2 * radii, values).astype(np.float64)
    img = np.where(beads > 0, beads + img, img)
    img = np.rint(img + rng.normal(0, read_noise, size=shape))
    return np.clip(img, 0, 65535).astype(np.uint16), np.column_stack([pos, radii])


def vignette(shape, strength=0.3, dtype=np.float32):
    h, w = shape
    yy, xx = np.mgrid[0:h, 0:w]
    rho2 = ((yy - (h - 1) / 2) / (h / 2)) ** 2 + ((xx - (w - 1) / 2) / (w / 2)) ** 2
    return (1 - strength * rho2 / 2).astype(dtype)
