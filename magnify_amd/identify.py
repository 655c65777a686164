"""identify_buttons (reference: src/magnify/identify.py:13-47).  identify_mrbles (spectral
decoding, identify.py:50-234) is outside the hot path; only its
ROI-reduce expression (identify.py:76-80) is, and that lives in ``magnify_amd.reduce``."""
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
def identify_mrbles(assay, spectra, codes, reference="eu"):
    raise NotImplementedError("identify_mrbles (lanthanide decoding) is outside the MI355X hot path; "
                              "use magnify_amd.reduce.fg_mean_minus_bg_median for its ROI-reduce step")
