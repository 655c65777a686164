"""identify_buttons (reference: src/magnify/identify.py:13-47) and the front half of
identify_mrbles (identify.py:50-90: intensities, lanthanide volumes and ratios); the spectral code
assignment (identify.py:92-234) is outside the hot path."""
from __future__ import annotations

import re

import numpy as np

from . import registry


@registry.component("identify_buttons")
def identify_buttons(assay, shape=None, pinlist=None, blank=None):
    if blank is None:
        blank = ["", "blank", "BLANK"]
    if pinlist is not None:
        import pandas as pd

        df = pd.read_csv(pinlist)
        df["Indices"] = df["Indices"].apply(lambda s: [int(x) for x in re.sub(r"[\(\)]", "", s).split(",")])
        df["MutantID"] = df["MutantID"].replace(blank, "")
        cols, rows = np.array(df["Indices"].to_list()).T - 1
        names = df["MutantID"].to_numpy(dtype=str, na_value="")
        names_array = np.empty((max(rows) + 1, max(cols) + 1), dtype=names.dtype)
        names_array[rows, cols] = names
    elif shape is not None:
        names_array = np.empty((shape[0], shape[1]), dtype="<U200")
        names_array.fill("default")
    else:
        raise ValueError("Either pinlist or shape must be provided.")
    n_t = assay.sizes["time"]
    return assay.assign_coords(
        tag=(("mark_row", "mark_col"), names_array),
        valid=(("mark_row", "mark_col", "time"), np.ones(names_array.shape + (n_t,), dtype=bool)),
    )


@registry.component("identify_mrbles")
def identify_mrbles(assay, spectra, codes, reference="eu", decode=False):
    """Front half of the reference's identify_mrbles (identify.py:50-90): per-bead
    intensities ``fg mean - bg median`` at time 0 (the fused device reductions of
    ``magnify_amd.reduce``), lanthanide volumes by least squares against the reference spectra
    (``S V = I``) and their ratios to the reference lanthanide, stored as ``ln_vol`` / ``ln_ratio``
    over the new ``ln`` coordinate.  Same argument meaning and ValueErrors as the reference
    (unknown reference lanthanide; lanthanide names of the two CSV files differ).

    The code-assignment back half (outlier removal, affine fit of the code grid, Gaussian-mixture
    assignment, identify.py:92-234) is outside this build: ``decode=True`` raises
    NotImplementedError after the volumes have been computed."""
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
    inten = reduce.fg_mean_minus_bg_median(assay).transpose("mark", "channel", "time").data
    inten = inten[:, use, 0].cpu().numpy()  # (mark, channel) at time 0
    volumes = np.linalg.lstsq(sp.T, inten.T, rcond=None)[0].T
    with np.errstate(invalid="ignore", divide="ignore"):
        ratios = volumes / volumes[:, 0:1]
    assay = assay.assign_coords(ln=("ln", lanthanides))
    assay["ln_vol"] = (("mark", "ln"), volumes)
    assay["ln_ratio"] = (("mark", "ln"), ratios)
    if decode:
        raise NotImplementedError("code assignment of identify_mrbles (identify.py:92-234) is not part of this build")
    return assay
