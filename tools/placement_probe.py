"""Do the two per-process levels of the correction pass and of the ROI pass show up between
BLOCKS of one process?  Several image blocks / ROI output sets are allocated side by side and the same pass is timed
into each (HIP events, best of 3).  python tools/placement_probe.py [--tries 4]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from magnify_amd import hotpath as hp  # noqa: E402
from magnify_amd.stack import StackProcessor, synthetic_stack  # noqa: E402
from synth import vignette  # noqa: E402


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tries", type=int, default=4)
    args = ap.parse_args()
    T, C, S = 64, 4, 4096
    stack, _ = synthetic_stack(T, C, S, S, seed=4000)
    flat = torch.from_numpy(vignette((S, S))).cuda()
    proc = StackProcessor(T, C, S, S, num_iter=5_000_000, search_channels=(0,), mode="P")
    out = proc(stack, flat, 100.0, seed=1)
    out = proc(stack, flat, 100.0, seed=2)
    torch.cuda.synchronize()
    tiles = stack.view(T * C, 1, 1, 1, S, S)
    res = {"apply_ms": [], "roi_ms": []}
    images = [proc.image] + [torch.empty_like(proc.image) for _ in range(args.tries - 1)]
    for img in images:
        res["apply_ms"].append(round(timed(lambda: hp.flatfield_stitch(tiles, 0, flat, 100.0, out=img, minmax_out=proc.minmax, n_groups=T)), 3))
    beads = out["beads"]
    for k in range(args.tries):
        res["roi_ms"].append(round(timed(lambda: hp.roi_gather_reduce(proc.image.view(T, C, 1, S, S), beads, proc.L, None, want_roi=True,
                                                                     reuse_buffers=True, disks=True, pool_tag=f"#probe{k}")), 3))
    # the same ROI outputs, the image block varied
    res["roi_ms_by_image"] = []
    for img in images:
        img.copy_(proc.image)
        res["roi_ms_by_image"].append(round(timed(lambda: hp.roi_gather_reduce(img.view(T, C, 1, S, S), beads, proc.L, None, want_roi=True,
                                                                               reuse_buffers=True, disks=True, pool_tag="#probe0")), 3))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
