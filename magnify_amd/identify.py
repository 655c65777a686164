"""identify_buttons (reference: src/magnify/identify.py:13-47) and identify_mrbles (identify.py:50-234:
intensities, lanthanide volumes and ratios from the device reductions; the spectral code assignment on
the resulting (mark, lanthanide) table is host-side NumPy / SciPy as in the reference)."""
from __future__ import annotations

import re

import numpy as np

from . import registry


@registry.component("identify_buttons")
def identify_buttons(assay, shape=None, pinlist=None, blank=None):
    """identify.py:13-47: the (mark_row, mark_col) table of tags -- read from a pin list (columns
    ``Indices`` = "(col, row)", 1-based, and ``MutantID``; names in ``blank`` become "") or, without one,
    "default" everywhere on a chip of ``shape`` -- plus an all-true ``valid`` over (mark_row, mark_col, time)."""
    blanks = ["", "blank", "BLANK"] if blank is None else blank
    if pinlist is not None:
        import pandas as pd

        table = pd.read_csv(pinlist)
        where = np.array([[int(v) for v in re.findall(r"-?\d+", str(text))] for text in table["Indices"]]) - 1
        labels = table["MutantID"].replace(blanks, "").to_numpy(dtype=str, na_value="")
        tags = np.empty((where[:, 1].max() + 1, where[:, 0].max() + 1), dtype=labels.dtype)
        tags[where[:, 1], where[:, 0]] = labels
    elif shape is not None:
        tags = np.full((shape[0], shape[1]), "default", dtype="<U200")
    else:
        raise ValueError("Either pinlist or shape must be provided.")
    alive = np.ones(tags.shape + (assay.sizes["time"],), dtype=bool)
    return assay.assign_coords(tag=(("mark_row", "mark_col"), tags), valid=(("mark_row", "mark_col", "time"), alive))


@registry.component("identify_mrbles")
def identify_mrbles(assay, spectra, codes, reference="eu", decode=True):
    """Front half of the reference's identify_mrbles (identify.py:50-90): per-bead
    intensities ``fg mean - bg median`` at time 0 (the fused device reductions of
    ``magnify_amd.reduce``), lanthanide volumes by least squares against the reference spectra
    (``S V = I``) and their ratios to the reference lanthanide, stored as ``ln_vol`` / ``ln_ratio``
    over the new ``ln`` coordinate.  Same argument meaning and ValueErrors as the reference
    (unknown reference lanthanide; lanthanide names of the two CSV files differ).

    ``decode`` (default, as the reference always does) adds the reference's code assignment (identify.py:88-234: outlier trimming, affine
    fit of the code levels, Gaussian-mixture assignment -- ``assign_codes``) as the ``tag`` coordinate over
    ``mark`` ("outlier" for beads the mixture gives to its uniform component); host-side NumPy / SciPy on the
    (mark, lanthanide) table, like the reference."""
    import pandas as pd

    from . import reduce

    table = pd.read_csv(spectra)
    hits = table.index[table["name"] == reference]
    if len(hits) == 0:
        raise ValueError(f"Reference lanthanide '{reference}' not found in spectra file")
    order = [hits[0]] + [i for i in range(len(table)) if i != hits[0]]  # the reference lanthanide first
    table = table.reindex(order)
    lanthanides = table["name"].to_list()
    code_table = pd.read_csv(codes)
    if set(code_table.columns) - {"name"} != set(lanthanides):
        raise ValueError(f"Lanthanide names in {codes} do not match lanthanide names in {spectra}.")

    names = [str(c) for c in np.asarray(assay.coords["channel"].values).tolist()]
    use = [i for i, c in enumerate(names) if c in table.columns]
    sp = table[[names[i] for i in use]].to_numpy(dtype=np.float64)  # (lanthanide, channel)
    inten = reduce.fg_mean_minus_bg_median(assay, time=0).data  # `assay.roi.isel(time=0)` first (identify.py:76)
    inten = inten[:, use, 0].cpu().numpy()  # (mark, channel) at time 0
    volumes = np.linalg.lstsq(sp.T, inten.T, rcond=None)[0].T
    with np.errstate(invalid="ignore", divide="ignore"):
        ratios = volumes / volumes[:, 0:1]
    assay = assay.assign_coords(ln=("ln", lanthanides))
    assay["ln_vol"] = (("mark", "ln"), volumes)
    assay["ln_ratio"] = (("mark", "ln"), ratios)
    if decode:
        code_ratios = code_table[lanthanides[1:]].to_numpy(dtype=np.float64)
        names = np.append(code_table["name"].to_numpy(), "outlier")
        tags, _, _ = assign_codes(ratios, code_ratios)
        assay = assay.assign_coords(tag=("mark", names[tags].astype(str)))
    return assay


def _fit_levels(points, levels, counts, n_grid=100):
    """identify.py:106-146: scale a and offset p that best lay the code levels over the sorted bead ratios,
    searched on an n_grid x n_grid lattice (a within +-25 % of the range ratio, p in the lowest quarter of
    the data range).  All lattice points are evaluated at once; a cluster's squared deviations come from
    prefix sums.  The reference's segment conventions are kept: a segment runs up to AND including the first
    point beyond the midpoint to the next level, which is also where the next segment starts."""
    if len(levels) == 1:
        return 1.0, float(points.mean())
    n, k = len(points), len(levels)
    scale = (points.max() - points.min()) / (levels.max() - levels.min())
    a = np.repeat(np.linspace(0.75 * scale, 1.25 * scale, n_grid), n_grid)
    p = np.tile(np.linspace(points.min(), 0.25 * points.max() + 0.75 * points.min(), n_grid), n_grid)
    clusters = a[:, None] * levels[None, :] + p[:, None]  # (grid, level)
    s1 = np.concatenate([[0.0], np.cumsum(points)])
    s2 = np.concatenate([[0.0], np.cumsum(points * points)])
    start = np.zeros(len(a), dtype=np.int64)
    dists = np.empty((len(a), k))
    sizes = np.empty((len(a), k))
    for i in range(k):
        if i < k - 1:
            first_beyond = np.searchsorted(points, (clusters[:, i] + clusters[:, i + 1]) / 2, side="right")
        else:
            first_beyond = np.full(len(a), n)
        j = np.where(first_beyond < n, np.maximum(first_beyond, start), n - 1)
        cnt = j + 1 - start
        c = clusters[:, i]
        sq = (s2[j + 1] - s2[start]) - 2.0 * c * (s1[j + 1] - s1[start]) + cnt * c * c
        dists[:, i] = np.where(j == start, np.inf, sq / cnt)
        sizes[:, i] = j - start
        start = j
    with np.errstate(invalid="ignore", divide="ignore"):
        share = sizes / sizes.sum(axis=1, keepdims=True) - (counts / counts.sum())[None, :]
        cost = 100 * dists.mean(axis=1) + (share ** 2).mean(axis=1)
    cost = np.where(np.isnan(cost), np.inf, cost)  # `cost < best` is false for NaN in the reference's loop
    best = int(np.argmin(cost))                   # first minimum, a outer / p inner as there
    if not np.isfinite(cost[best]):
        return 0.0, 0.0
    return float(a[best]), float(p[best])


def assign_codes(ratios, code_ratios, n_grid=100, em_steps=50):
    """Code assignment of identify_mrbles (identify.py:88-228).  ``ratios`` (bead, lanthanide) with the
    reference lanthanide in column 0; ``code_ratios`` (code, lanthanide - 1).  Returns (tag index per bead
    -- ``len(codes)`` = outlier --, A, p).
      1. trim: beads whose distance to their n-th neighbour is in the top 5 % are set aside;
      2. per lanthanide, fit the code levels to the trimmed ratios (``_fit_levels``); label by nearest code;
      3. 50 EM steps of a Gaussian mixture (one shared initial covariance) plus a uniform outlier component,
         evaluated on all beads; label = most probable component."""
    import scipy.spatial
    import scipy.special

    X = np.asarray(ratios, dtype=np.float64)[:, 1:]
    code_ratios = np.asarray(code_ratios, dtype=np.float64)
    n_codes, dims = code_ratios.shape
    nth = round(len(X) / (20 * n_codes)) + 2
    reach = scipy.spatial.KDTree(X, leafsize=nth).query(X, k=[nth])[0].ravel()
    core = X[reach <= np.percentile(reach, 95)]
    A, p = np.zeros(dims), np.zeros(dims)
    for d in range(dims):
        levels, counts = np.unique(code_ratios[:, d], return_counts=True)
        A[d], p[d] = _fit_levels(np.sort(core[:, d]), levels, counts, n_grid)
    centres = A * code_ratios + p
    nearest = np.argmin(np.linalg.norm(core[:, None] - centres[None], axis=-1), axis=1)
    means = np.zeros((n_codes, dims))
    covs = np.zeros((n_codes, dims, dims)) + np.eye(dims) * 1e-10
    weight = np.zeros(n_codes + 1)
    for k in range(n_codes):
        own = core[nearest == k]
        weight[k] = len(own) + 1
        with np.errstate(invalid="ignore"), np.testing.suppress_warnings() as quiet:
            quiet.filter(RuntimeWarning)
            means[k] = np.median(own, axis=0)
            if len(own) > 0:
                covs[k] += np.cov(own, rowvar=False)
    covs[:] = np.median(covs, axis=0)
    weight[-1] = 1e-10
    weight /= weight.sum()
    log_like = np.empty((len(X), n_codes + 1))
    log_like[:, -1] = -np.log(core.max(axis=0) - core.min(axis=0)).sum()  # uniform over the trimmed range
    resp = None
    for _ in range(em_steps):
        off = X[:, None, :] - means[None]
        try:
            log_like[:, :-1] = (-dims * np.log(2 * np.pi) / 2 - 0.5 * np.log(np.linalg.det(covs))
                                - 0.5 * np.einsum("...i,...ij,...j->...", off, np.linalg.inv(covs), off))
        except np.linalg.LinAlgError:
            print("Warning: Code clustering did not converge.")
            break
        log_post = np.log(weight) + log_like
        log_post -= scipy.special.logsumexp(log_post, axis=1)[:, None]
        resp = np.exp(log_post)
        mass = resp[:, :-1].sum(axis=0)
        means = (resp[:, :-1, None] * X[:, None, :]).sum(axis=0) / mass[:, None]
        off = X[:, None, :] - means[None]
        covs = (resp[:, :-1, None, None] * np.einsum("...i,...j->...ij", off, off)).sum(axis=0) / mass[:, None, None]
        covs += np.eye(dims) * np.median(covs) / 10
        weight = resp.sum(axis=0) / len(X)
    if resp is not None:
        tags = np.argmax(resp, axis=1)
    else:
        tags = np.argmin(np.linalg.norm(X[:, None] - centres[None], axis=-1), axis=1)
    return tags, A, p
