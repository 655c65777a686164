"""What limits the flat-field correction pass: the same launch (64 x 4 planes of 4096^2 uint16, same workgroups) as a
plain crop-and-copy (no arithmetic), with / without the min/max, beside the correction itself.
python tools/apply_probe.py [--timepoints 64]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from magnify_amd import hotpath as hp  # noqa: E402
from synth import vignette  # noqa: E402


def timed(fn, reps=6):
    fn()
    torch.cuda.synchronize()
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
    ap.add_argument("--timepoints", type=int, default=64)
    args = ap.parse_args()
    T, C, n = args.timepoints, 4, 4096
    g = torch.Generator(device="cuda").manual_seed(1)
    tiles = torch.randint(100, 4000, (T, C, 1, 1, n, n), dtype=torch.int32, device="cuda", generator=g).to(torch.uint16)
    out = torch.empty((T, C, n, n), dtype=torch.uint16, device="cuda")
    mm = torch.empty((T * C, 2), dtype=torch.float64, device="cuda")
    flat = torch.from_numpy(vignette((n, n))).cuda()
    max2 = hp.flatfield_max(tiles, flat, 100.0, T)
    gb = 2 * tiles.numel() * 2 / 1e9
    for label, kw in (("copy, no min/max", dict(apply_flatfield=False, want_minmax=False)),
                      ("copy + min/max", dict(apply_flatfield=False, want_minmax=True)),
                      ("correction, no min/max", dict(apply_flatfield=True, want_minmax=False)),
                      ("correction + min/max", dict(apply_flatfield=True, want_minmax=True))):
        ms = timed(lambda: hp.flatfield_stitch(tiles, 0, flat, 100.0, max2=max2, out=out, minmax_out=mm, n_groups=T, **kw))
        print(f"{label:26s} {ms:7.3f} ms  {gb / ms * 1e3:7.1f} GB/s")
    ms = timed(lambda: hp.flatfield_max(tiles, flat, 100.0, T))
    print(f"{'maxima pass':26s} {ms:7.3f} ms  {gb / 2 / ms * 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()
