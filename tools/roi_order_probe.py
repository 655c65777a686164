"""Does the ORDER in which the ROI pass visits an assay's windows matter (L2 reuse of the lines that neighbouring
windows share)?  Same beads, same work; the bead tables are permuted: as found (score order = spatially random),
sorted by (row band, column), and sorted + dealt so that every 8th workgroup (one XCD under round-robin dispatch)
walks a contiguous run of the sorted list.  python tools/roi_order_probe.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnify_amd import hotpath as hp  # noqa: E402
from magnify_amd.stack import StackProcessor, synthetic_stack  # noqa: E402

T, C, S = 64, 4, 4096
stack, _ = synthetic_stack(T, C, S, S, seed=4000)
proc = StackProcessor(T, C, S, S, num_iter=5_000_000, search_channels=(0,), mode="P")
proc.flatfield(stack, 1.0, 100.0)
beads = [np.asarray(b) for b in proc.detect(0)]
images = proc.image.view(T, C, 1, S, S)


def timeit(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return round(min(ts), 3)


def banded(b, band):
    return b[np.lexsort((b[:, 1], b[:, 0] // band))]


def dealt(b, groups=8):
    n = len(b)
    per = -(-n // groups)
    idx = np.arange(n)
    src = (idx % groups) * per + idx // groups
    src = src[src < n]
    rest = np.setdiff1d(np.arange(n), src, assume_unique=False)
    return b[np.concatenate([src, rest])]


res = {"markers": int(sum(len(b) for b in beads))}
variants = {"as_found": beads}
for band in (32, 64, 128, 256):
    variants["banded_%d" % band] = [banded(b, band) for b in beads]
    variants["banded_%d_dealt8" % band] = [dealt(banded(b, band)) for b in beads]
for name, bl in variants.items():
    res[name] = timeit(lambda: hp.roi_gather_reduce(images, bl, 100, None, want_roi=True, reuse_buffers=True, disks=True))
print(json.dumps(res))
