"""magnify_amd -- the marker-detection hot path of FordyceLab/magnify (tile stitch, flat-field,
bead/button finding, fg/bg segmentation, per-ROI reduction) rebuilt MI355X-native: hand-written
HIP kernels behind a C ABI, driven through magnify's own registry / Pipeline component API.

Public names mirror the reference's ``magnify/__init__.py:25-40``.
"""
__version__ = "0.1.0"

from . import xr_lite
from .xr_lite import DataArray, Dataset
from .registry import (  # noqa: F401
    beads,
    beads_pipe,
    component,
    components,
    image,
    image_pipe,
    microfluidic_chip,
    microfluidic_chip_pipe,
    mrbles,
    mrbles_pipe,
    readers,
)
from .pipeline import Pipeline  # noqa: F401
from . import reader, preprocess, stitch, find, identify, postprocess, reduce, utils, filter  # noqa: F401,E402  (register components)
from .utils import seed  # noqa: F401
from .file import load, save  # noqa: F401
from . import sink  # noqa: F401,E402
from .sink import HostSink, SaveSink  # noqa: F401,E402

__all__ = ["component", "microfluidic_chip", "microfluidic_chip_pipe", "mrbles", "mrbles_pipe", "beads", "beads_pipe",
           "image", "image_pipe", "save", "load", "Pipeline", "DataArray", "Dataset", "seed", "find", "identify", "postprocess",
           "preprocess", "reader", "stitch", "reduce", "utils", "xr_lite", "filter"]
