"""The ROI pass of the C4 step with and without mg_roi_window_order, alternating inside ONE process on the same blocks
(the pass has two per-process levels, DESIGN.md section 5: separate processes cannot tell 0.2 ms apart)."""
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
out = proc(stack, 1.0, 100.0, seed=1)
d_beads = out["device_tables"][0]
counts = [len(b) for b in out["beads"]]
images = proc.image.view(T, C, 1, S, S)


def once(flag):
    hp._ROI_ORDER = flag
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    hp.roi_gather_reduce(images, None, proc.L, None, want_roi=True, reuse_buffers=True, disks=True,
                         device_tables=(d_beads, counts, proc.max_r))
    b.record()
    b.synchronize()
    return a.elapsed_time(b)


res = {"ordered": [], "as_listed": []}
for flag in (True, False):
    once(flag)
for rep in range(6):
    res["ordered"].append(round(once(True), 3))
    res["as_listed"].append(round(once(False), 3))
print(json.dumps({"markers": int(sum(counts)), **res}))
