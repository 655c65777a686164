"""Mode R (the reference's single-assay semantics, find.py:477, 543-550: beads detected at time 0 and
propagated over time, flat-field maxima over the whole stack) on the C4 stack; SURVEY 8d asks for it beside
the mode-P headline of bench.py.

    python tools/mode_r_bench.py [--timepoints 64] [--steps 3]
"""
import argparse
import json
import os
import sys
import time

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))

import torch  # noqa: E402

from magnify_amd import hotpath as hp  # noqa: E402
from magnify_amd.stack import StackProcessor, synthetic_stack  # noqa: E402
from synth import vignette  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--timepoints", type=int, default=64)
ap.add_argument("--channels", type=int, default=4)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--num-iter", type=int, default=5_000_000)
args = ap.parse_args()
hp.require_gpu()
T, C, S = args.timepoints, args.channels, args.size
stack, _ = synthetic_stack(T, C, S, S, seed=4000, jitter=0)
flat = torch.from_numpy(vignette((S, S))).cuda()
proc = StackProcessor(T, C, S, S, num_iter=args.num_iter, search_channels=(0,), mode="R")
out = proc(stack, flat, 100.0, seed=0)
torch.cuda.synchronize()
timer = hp.StageTimer()
hp.set_timer(timer)
t0 = time.perf_counter()
for i in range(args.steps):
    out = proc(stack, flat, 100.0, seed=1 + i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.steps
hp.set_timer(None)
m = len(out["beads"][0])
L = proc.L
roi_bytes = m * T * L * L * (4 * C + 2)
print(json.dumps({"mode": "R", "shape": [T, C, S, S], "ms_per_step": dt * 1e3, "MPs": T * C * S * S / dt / 1e6,
                  "markers_at_t0": m, "roi_windows": m * T, "roi_windows_per_s": m * T / dt,
                  "roi_algorithmic_GB": roi_bytes / 1e9, "placement": proc.placement,
                  "stages_ms": {k: round(v[0] / args.steps, 3) for k, v in sorted(timer.summary().items(), key=lambda kv: -kv[1][0])}}))
