"""TEST INFRASTRUCTURE ONLY (only tests/, smoke() and bench.py's cpu_baseline may use oracle/).

Plain-loop restatement of the code assignment of the reference's identify_mrbles
(src/magnify/identify.py:88-234): outlier trimming by k-th neighbour distance, per-lanthanide
affine fit of the code levels on a 100 x 100 grid, nearest-code labelling, 50 EM steps of a Gaussian
mixture with a uniform outlier component.  PARITY UNPINNED: the reference has no test or golden
vector for this function; this file follows its source text statement by statement (including the
inclusive slice ends and the loop-variable carry-over of its 1-D fit)."""
import numpy as np
import scipy.spatial
import scipy.special


def fit_1d(points, codes, counts, n_grid=100):
    """identify.py:106-146.  points sorted ascending; codes ascending distinct levels; counts per level."""
    if len(codes) == 1:
        return 1.0, float(points.mean())
    scale = (points.max() - points.min()) / (codes.max() - codes.min())
    sizes = np.zeros(len(codes))
    dists = np.ones(len(codes))
    best_a, best_p, best_cost = 0.0, 0.0, np.inf
    for a in np.linspace(0.75 * scale, 1.25 * scale, n_grid):
        for p in np.linspace(points.min(), 0.25 * points.max() + 0.75 * points.min(), n_grid):
            clusters = a * codes + p
            start = 0
            j = 0  # numba leaves the loop variable at its last value when a range is empty
            for i in range(len(clusters)):
                mid = (clusters[i] + clusters[i + 1]) / 2 if i < len(clusters) - 1 else np.inf
                for j in range(start, len(points)):
                    if points[j] > mid:
                        break
                if start == j:
                    dists[i] = np.inf
                else:
                    dists[i] = ((points[start: j + 1] - clusters[i]) ** 2).mean()
                sizes[i] = j - start
                start = j
            with np.errstate(invalid="ignore", divide="ignore"):
                cost = 100 * dists.mean() + ((sizes / sizes.sum() - counts / counts.sum()) ** 2).mean()
            if cost < best_cost:
                best_a, best_p, best_cost = a, p, cost
    return best_a, best_p


def assign_codes(ratios, code_ratios, n_grid=100, em_steps=50):
    """identify.py:88-228 -> (index into the codes, or len(codes) for 'outlier', per bead; A; p)."""
    X = ratios[:, 1:]
    num_codes, dims = code_ratios.shape
    n_neighbor = round(len(X) / (20 * num_codes)) + 2
    dist = scipy.spatial.KDTree(X, leafsize=n_neighbor).query(X, k=[n_neighbor])[0].flatten()
    X_r = X[dist <= np.percentile(dist, 95)]
    A, p = np.zeros(dims), np.zeros(dims)
    for i in range(dims):
        c, counts = np.unique(code_ratios[:, i], return_counts=True)
        A[i], p[i] = fit_1d(np.sort(X_r[:, i]), c, counts, n_grid)
    centres = A * code_ratios + p
    idx = np.argmin(np.linalg.norm(X_r[:, None] - centres[None], axis=-1), axis=1)
    means = np.zeros((num_codes, dims))
    covs = np.zeros((num_codes, dims, dims)) + np.eye(dims) * 1e-10
    prop = np.zeros(num_codes + 1)
    for k in range(num_codes):
        prop[k] = np.sum(idx == k) + 1
        with np.errstate(invalid="ignore"), np.testing.suppress_warnings() as sup:
            sup.filter(RuntimeWarning)
            means[k] = np.median(X_r[idx == k], axis=0)
            if prop[k] > 1:
                covs[k] += np.cov(X_r[idx == k], rowvar=False)
    covs[:] = np.median(covs, axis=0)
    prop[-1] = 1e-10
    prop /= prop.sum()
    log_cond = np.empty((len(X), num_codes + 1))
    log_cond[:, -1] = -np.log(X_r.max(axis=0) - X_r.min(axis=0)).sum()
    probs = None
    for _ in range(em_steps):
        diff = X[:, None, :] - means[None]
        try:
            log_cond[:, :-1] = (-dims * np.log(2 * np.pi) / 2 - 0.5 * np.log(np.linalg.det(covs))
                                - 0.5 * np.einsum("...i,...ij,...j->...", diff, np.linalg.inv(covs), diff))
        except np.linalg.LinAlgError:
            break
        log_probs = np.log(prop) + log_cond
        log_probs -= scipy.special.logsumexp(log_probs, axis=1)[:, None]
        probs = np.exp(log_probs)
        means = np.sum(probs[:, :-1, None] * X[:, None, :], axis=0) / np.sum(probs[:, :-1], axis=0)[:, None]
        diff = X[:, None, :] - means[None]
        covs = (np.sum(probs[:, :-1, None, None] * np.einsum("...i,...j->...ij", diff, diff), axis=0)
                / np.sum(probs[:, :-1], axis=0)[:, None, None])
        covs += np.eye(dims) * np.median(covs) / 10
        prop = np.sum(probs, axis=0) / X.shape[0]
    if probs is not None:
        tags = np.argmax(probs, axis=1)
    else:
        tags = np.argmin(np.linalg.norm(X[:, None] - centres[None], axis=-1), axis=1)
    return tags, A, p
