"""Diagnostic: where does the host spend time in one StackProcessor step? (cProfile + wall clock)"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

from magnify_amd.stack import StackProcessor, synthetic_stack
from synth import vignette

T = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = 4096
dev = torch.device("cuda")
stack, _ = synthetic_stack(T, 4, S, S, device=dev)
flat = torch.from_numpy(vignette((S, S))).to(dev)
proc = StackProcessor(T, 4, S, S, num_iter=5_000_000, min_bead_diameter=10, max_bead_diameter=50, search_channels=(0,))
for i in range(2):
    proc(stack, flat, 100.0, seed=i)
torch.cuda.synchronize()
for i in range(2):
    t0 = time.perf_counter()
    proc.flatfield(stack, flat, 100.0); torch.cuda.synchronize(); t1 = time.perf_counter()
    beads = proc.detect(i); torch.cuda.synchronize(); t2 = time.perf_counter()
    out = proc.segment_reduce(beads); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"flatfield {1e3*(t1-t0):.1f} ms  detect {1e3*(t2-t1):.1f} ms  segment_reduce {1e3*(t3-t2):.1f} ms")
pr = cProfile.Profile()
pr.enable()
proc(stack, flat, 100.0, seed=5)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
