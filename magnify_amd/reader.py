"""Reader (reference: src/magnify/reader.py:23-77).  Only the in-memory pass-through branch
(reader.py:31-35) is on the hot path and implemented: DataArray / Dataset objects (magnify_amd's or
real xarray's) or a sequence of them are yielded one assay at a time.  The path-pattern parser and
the lazy TIFF reader (reader.py:80-324) are file I/O, out of scope for this build.
"""
from __future__ import annotations

import os

from . import registry, xr_lite


class Reader:
    def __call__(self, data):
        single = isinstance(data, (str, bytes, os.PathLike, xr_lite.DataArray, xr_lite.Dataset)) or \
            type(data).__module__.startswith("xarray")
        for d in ([data] if single else data):
            if isinstance(d, (str, bytes, os.PathLike)):
                if not os.path.exists(os.fspath(d)) and not any(ch in str(d) for ch in "*?({"):
                    raise FileNotFoundError(f"The pattern {d} did not lead to any files.")
                raise NotImplementedError("reading image files is outside the MI355X hot path (SURVEY.md 8f, N2); "
                                          "pass an in-memory DataArray")
            yield xr_lite.from_any(d)

    @registry.readers.register("read")
    def make():
        return Reader()
