"""Config C5 rehearsal from host memory: a TILED time series (T x C x rows x cols tiles of tile^2 with overlap)
arrives chunk by chunk from pinned host buffers; `stack.process_stream` uploads a chunk, crops / joins / corrects
the tiles on the device and runs the hot path; the stitched assay never exists on the host.  PCIe-inclusive
throughput in stitched megapixels per second (the figure that is NOT bench.py's `value`).

    python tools/c5_stream_bench.py [--timepoints 32] [--chunk 8] [--grid 4] [--tile 1126] [--overlap 102] [--streams 1]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from magnify_amd.stack import process_stream, stitched_shape, synthetic_stack  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timepoints", type=int, default=32)
    ap.add_argument("--chunk", type=int, default=8)
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--grid", type=int, default=4)
    ap.add_argument("--tile", type=int, default=1126)
    ap.add_argument("--overlap", type=int, default=102)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--num-iter", type=int, default=5_000_000)
    args = ap.parse_args()
    T, C, R, ty, ov = args.timepoints, args.channels, args.grid, args.tile, args.overlap
    h, w = stitched_shape(R, R, ty, ty, ov)
    clip, rem = ov // 2, ov % 2
    step = ty - 2 * clip - rem
    # a canvas large enough to cut overlapping tiles from: tile (r, c) starts at (r * step, c * step)
    side = (R - 1) * step + ty
    chunks = []
    for t0 in range(0, T, args.chunk):
        n = min(args.chunk, T - t0)
        canvas, _ = synthetic_stack(n, C, side, side, seed=5000 + t0)
        tiles = torch.empty((n, C, R, R, ty, ty), dtype=torch.uint16, device="cuda")
        for r in range(R):
            for c in range(R):
                tiles[:, :, r, c] = canvas[:, :, r * step:r * step + ty, c * step:c * step + ty]
        chunks.append(tiles.cpu().pin_memory())
        del canvas, tiles
    torch.cuda.synchronize()
    yy, xx = np.mgrid[0:ty, 0:ty]
    flat = (1 - 0.15 * (((yy - (ty - 1) / 2) / (ty / 2)) ** 2 + ((xx - (ty - 1) / 2) / (ty / 2)) ** 2)).astype(np.float32)
    kw = dict(num_iter=args.num_iter, search_channels=(0,), n_streams=args.streams)

    def run():
        markers = 0
        for out in process_stream(iter(chunks), flat, 100.0, seed=7, overlap=ov, **kw):
            markers += sum(len(b) for b in out["beads"])
        torch.cuda.synchronize()
        return markers

    run()  # workspaces
    t0 = time.perf_counter()
    markers = run()
    dt = time.perf_counter() - t0
    tile_bytes = T * C * R * R * ty * ty * 2
    print(json.dumps({"workload": f"C5 rehearsal: {T} timepoints x {C} ch x {R}x{R} tiles of {ty}^2 (overlap {ov} -> {h}x{w}), "
                                  f"chunks of {args.chunk} timepoints from pinned host memory, num_iter={args.num_iter}",
                      "streams": args.streams, "seconds": dt, "ms_per_timepoint": 1e3 * dt / T,
                      "stitched_MPs": T * C * h * w / dt / 1e6, "host_to_device_GBs": tile_bytes / dt / 1e9,
                      "markers": markers, "markers_per_s": markers / dt}))


if __name__ == "__main__":
    main()
