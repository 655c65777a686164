"""How much would two C4 steps in flight at once gain?  Two StackProcessors, each with its own HIP stream and host
thread, work through steps of the same stack side by side (their kernels meet on the GPU: the bandwidth-bound passes of
one beside the vector- / LDS-bound chain of the other); against one processor doing the same number of steps alone.
python tools/pipeline_probe.py [--steps 10]"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MG_PLACEMENT_TRIES", "0")

import torch  # noqa: E402

from magnify_amd.stack import StackProcessor, synthetic_stack  # noqa: E402
from synth import vignette  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--timepoints", type=int, default=64)
    ap.add_argument("--lanes", type=int, default=2)
    args = ap.parse_args()
    T, C, S = args.timepoints, 4, 4096
    stack, _ = synthetic_stack(T, C, S, S, seed=4000)
    flat = torch.from_numpy(vignette((S, S))).cuda()
    procs, streams = [], []
    for k in range(args.lanes):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            p = StackProcessor(T, C, S, S, num_iter=5_000_000, search_channels=(0,), mode="P")
            p.pool_tag = "#lane%d" % k
            for i in range(5):  # warm-up: captures happen here, one lane at a time
                p(stack, flat, 100.0, seed=100 * k + i)
        s.synchronize()
        procs.append(p)
        streams.append(s)
    torch.cuda.synchronize()

    def run(k, n, counts):
        with torch.cuda.stream(streams[k]):
            for i in range(n):
                out = procs[k](stack, flat, 100.0, seed=1000 * (k + 1) + i)
                counts[k] += sum(len(b) for b in out["beads"])
            streams[k].synchronize()

    res = {}
    counts = [0] * args.lanes
    t0 = time.perf_counter()
    run(0, args.steps, counts)
    res["one_lane_ms_per_step"] = 1e3 * (time.perf_counter() - t0) / args.steps
    counts = [0] * args.lanes
    threads = [threading.Thread(target=run, args=(k, args.steps, counts)) for k in range(args.lanes)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    res["%d_lanes_ms_per_step" % args.lanes] = 1e3 * (time.perf_counter() - t0) / (args.lanes * args.steps)
    res["markers_per_step"] = sum(counts) / (args.lanes * args.steps)
    res["graph_replays"] = [getattr(p.finder, "graph_replays", None) for p in procs]
    print(json.dumps(res))


if __name__ == "__main__":
    main()
