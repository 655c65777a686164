"""Where a single-assay call (BASELINE config C2: 1 x 4 x 4096^2 through mg.beads) spends its time:
wall clock per call, the host-side profile (cProfile) and -- when run under
`rocprofv3 --kernel-trace --stats` -- the kernels.  Strong scaling hangs on this.

    python tools/c2_timeline.py [--calls 30] [--size 4096] [--channels 4] [--profile]
"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import magnify_amd as mg  # noqa: E402
from magnify_amd.stack import synthetic_stack  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=30)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--num-iter", type=int, default=5_000_000)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--stages", action="store_true", help="HIP-event time of every C-ABI call, averaged over the calls")
    ap.add_argument("--loop-iters", type=int, default=20)
    ap.add_argument("--loops", action="store_true", help="back-to-back loops of the two flat-field kernels")
    args = ap.parse_args()
    stack, _ = synthetic_stack(1, args.channels, args.size, args.size, seed=2000)
    planes = stack[0]

    def run():
        mg.seed(2100)
        return mg.beads(data=mg.DataArray(planes, ("channel", "y", "x")), overlap=0, num_iter=args.num_iter, search_channel=0)

    for _ in range(3):
        xp = run()
    torch.cuda.synchronize()
    times = []
    for _ in range(args.calls):
        t0 = time.perf_counter()
        xp = run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times.sort()
    print(f"markers {xp.roi.sizes['mark']}  best {1e3 * times[0]:.3f} ms  median {1e3 * times[len(times) // 2]:.3f} ms")
    if args.stages:
        from magnify_amd import hotpath

        timer = hotpath.StageTimer()
        hotpath.set_timer(timer)
        for _ in range(args.calls):
            run()
        hotpath.set_timer(None)
        for name, (tot, cnt) in sorted(timer.summary().items(), key=lambda kv: -kv[1][0]):
            print(f"  {name:28s} {1e3 * tot / args.calls:8.1f} us/call  {cnt / args.calls:5.1f} launches/call")
    if args.loops:
        from magnify_amd import hotpath

        tiles = planes.reshape(args.channels, 1, 1, 1, args.size, args.size)
        for label, fn in (("flatfield_max", lambda: hotpath.flatfield_max(tiles, 1.0, 0.0)),
                          ("flatfield_stitch", lambda: hotpath.flatfield_stitch(tiles, 0, 1.0, 0.0, max2=m2))):
            m2 = hotpath.flatfield_max(tiles, 1.0, 0.0)
            fn()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.loop_iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            print(f"  {label}: {1e3 * a.elapsed_time(b) / args.loop_iters:.1f} us per back-to-back call")
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.calls):
            run()
            torch.cuda.synchronize()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(70)
        print(s.getvalue())


if __name__ == "__main__":
    main()
