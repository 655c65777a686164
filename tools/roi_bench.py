"""Micro-benchmark of the ROI stage alone: label-map masks vs disk masks on the same beads."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from magnify_amd import hotpath as hp  # noqa: E402
from magnify_amd.stack import StackProcessor, synthetic_stack  # noqa: E402

T, C, S = 16, 4, 4096
stack, _ = synthetic_stack(T, C, S, S, seed=4000)
proc = StackProcessor(T, C, S, S, num_iter=5_000_000, search_channels=(0,), mode="P")
proc.flatfield(stack, 1.0, 100.0)
beads = proc.detect(0)
images = proc.image.view(T, C, 1, S, S)
print("markers", sum(len(b) for b in beads))


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


labels = hp.circle_labels(beads, S, S)
t_lab = timeit(lambda: hp.roi_gather_reduce(images, beads, 100, labels, reuse_buffers=True))
t_disk = timeit(lambda: hp.roi_gather_reduce(images, beads, 100, None, reuse_buffers=True, disks=True))
t_none = timeit(lambda: hp.roi_gather_reduce(images, beads, 100, None, reuse_buffers=True))
print(f"labels {t_lab:.3f} ms  disks {t_disk:.3f} ms  no-masks {t_none:.3f} ms  (host+device, {T} assays)")
