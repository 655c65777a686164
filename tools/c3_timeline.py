"""Where a C3 call (8 x 8 tiles of 1024^2 -> 7376^2, 28 x 28 buttons, through mg.microfluidic_chip) spends its time:
HIP-event time per C-ABI stage and the host profile.  python tools/c3_timeline.py [--calls 10] [--profile]"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import magnify_amd as mg  # noqa: E402
from magnify_amd import hotpath  # noqa: E402
from synth import draw_chip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=10)
    ap.add_argument("--profile", action="store_true")
    args = ap.parse_args()
    n, pitch, ty, overlap = 28, 250, 1024, 102
    canvas = draw_chip((n, n), 20, row_dist=pitch, col_dist=pitch)
    step = ty - overlap
    need = 7 * step + ty
    big = np.zeros((max(need, canvas.shape[0]), max(need, canvas.shape[1])), dtype=np.uint16)
    big[: canvas.shape[0], : canvas.shape[1]] = canvas
    tiles = np.stack([np.stack([big[r * step: r * step + ty, c * step: c * step + ty] for c in range(8)]) for r in range(8)])
    dev = torch.from_numpy(tiles).cuda()

    def run():
        mg.seed(3000)
        return mg.microfluidic_chip(data=mg.DataArray(dev, ("row", "col", "y", "x")), shape=(n, n), overlap=overlap,
                                    row_dist=pitch, col_dist=pitch, min_button_diameter=8, max_button_diameter=30)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    times = []
    for _ in range(args.calls):
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times.sort()
    print(f"best {1e3 * times[0]:.3f} ms  median {1e3 * times[len(times) // 2]:.3f} ms")
    timer = hotpath.StageTimer()
    hotpath.set_timer(timer)
    for _ in range(args.calls):
        run()
    hotpath.set_timer(None)
    tot = 0.0
    for name, (ms, cnt) in sorted(timer.summary().items(), key=lambda kv: -kv[1][0]):
        tot += ms / args.calls
        print(f"  {name:28s} {1e3 * ms / args.calls:8.1f} us/call  {cnt / args.calls:5.1f} launches/call")
    print(f"  stages together {tot:.3f} ms/call")
    from magnify_amd import find as mgfind

    for key, f in mgfind._FINDERS.items():
        print(f"  finder P={f.P} {f.h}x{f.w}: graph replays {f.graph_replays}, captures {f.graph_captures}, "
              f"error {f.stats.get('graph_error')}")
    if args.profile:
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(args.calls):
            run()
            torch.cuda.synchronize()
        pr.disable()
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
        print(s.getvalue())


if __name__ == "__main__":
    main()
