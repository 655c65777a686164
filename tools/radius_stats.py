"""Distribution of the unique circles over radii / 64 x 64 tiles on the bench workload (one plane):
how full the one-radius chunks of the scoring prefilter are.  usage: python tools/radius_stats.py"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from magnify_amd.stack import StackProcessor, synthetic_stack
from magnify_amd import _native as nat
from synth import vignette
T, C, S = 2, 4, 4096
stack, _ = synthetic_stack(T, C, S, S, seed=4000)
flat = torch.from_numpy(vignette((S, S))).cuda()
proc = StackProcessor(T, C, S, S, num_iter=5_000_000, search_channels=(0,), mode="P")
proc(stack, flat, 100.0, seed=1)
f = proc.finder
ls = f.layer_starts[0].cpu().numpy().astype(np.int64)        # (n_tiles, nr + 1)
cnt = np.diff(ls, axis=1)                                       # (n_tiles, nr)
ntr = ntc = int(round(np.sqrt(cnt.shape[0])))
per = np.diff(f.per_starts.cpu().numpy())
print("circles per radius (plane 0):", cnt.sum(axis=0).tolist())
print("perimeter lengths:", per.tolist())
tot = cnt.sum()
print("weighted mean perimeter:", float((cnt.sum(axis=0) * per).sum() / tot))
c4 = cnt.reshape(ntr, ntc, -1)
for name, (a, b) in {"64x64": (1, 1), "128x128": (2, 2), "128x256": (2, 4), "256x256": (4, 4)}.items():
    pr, pc = (-ntr) % a, (-ntc) % b
    cc = np.pad(c4, ((0, pr), (0, pc), (0, 0)))
    cc = cc.reshape(cc.shape[0] // a, a, cc.shape[1] // b, b, -1).sum(axis=(1, 3))
    chunks = (cc + 63) // 64
    work = (chunks * per).sum()          # wave-reads
    useful = (cc * per).sum() / 64
    half = (((cc + 31) // 32) * per).sum() / 2
    print(f"{name}: super-tiles {cc.shape[0] * cc.shape[1]}, wave-reads {work}, useful {useful:.0f}, utilisation {useful / work:.2f}, "
          f"with 32-lane granularity {useful / half:.2f}")
