"""Config C5 measured: a TILED time series (T x C x rows x cols tiles of tile^2 with overlap) arrives chunk by chunk
-- from pinned host buffers, or (--files DIR) from OME-BigTIFF files, one per tile position with (time, channel) pages,
read page by page through magnify_amd.tiff / reader.iter_time_chunks --; `stack.process_stream` uploads a chunk,
crops / joins / corrects the tiles on the device and runs the hot path; the stitched assay never exists on the host.
--sink host|save keeps / writes every timepoint's results (magnify_amd.sink).  PCIe- (and file-) inclusive throughput in
stitched megapixels per second (the figure that is NOT bench.py's `value`).

    python tools/c5_stream_bench.py [--timepoints 64] [--chunk 8] [--grid 4] [--tile 1126] [--overlap 102] [--files DIR]
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
    ap.add_argument("--timepoints", type=int, default=64)
    ap.add_argument("--files", default=None, help="write the series as OME-BigTIFF files into this directory and stream it from there")
    ap.add_argument("--sink", choices=("none", "host", "save"), default="none")
    ap.add_argument("--want-roi", action="store_true", help="ROI pixel stacks and masks travel to the sink as well")
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
    write_s = None
    if args.files:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from tiffwrite import ome_xml, write_tiff

        os.makedirs(args.files, exist_ok=True)
        t0 = time.perf_counter()
        for r in range(R):
            for c in range(R):
                pages = [chunks[t // args.chunk][t % args.chunk, ch, r, c].numpy() for t in range(T) for ch in range(C)]
                write_tiff(os.path.join(args.files, f"acq_r{r}_c{c}.ome.tif"), pages, bigtiff=True,
                           description=ome_xml(size_c=C, size_t=T, size_y=ty, size_x=ty))
        write_s = time.perf_counter() - t0
        chunks = None  # from here on the series only exists in the files
    yy, xx = np.mgrid[0:ty, 0:ty]
    flat = (1 - 0.15 * (((yy - (ty - 1) / 2) / (ty / 2)) ** 2 + ((xx - (ty - 1) / 2) / (ty / 2)) ** 2)).astype(np.float32)
    kw = dict(num_iter=args.num_iter, search_channels=(0,), n_streams=args.streams)

    def source():
        if args.files:
            from magnify_amd import reader

            return reader.iter_time_chunks(os.path.join(args.files, "acq_r(row)_c(col).ome.tif"), args.chunk, pinned=True)
        return iter(chunks)

    def run():
        import tempfile

        import magnify_amd as mg

        markers, sink, tmp = 0, None, None
        if args.sink == "host":
            sink = mg.HostSink(want_roi=args.want_roi, want_masks=args.want_roi)
        elif args.sink == "save":
            tmp = tempfile.TemporaryDirectory(prefix="c5_results_")
            sink = mg.SaveSink(os.path.join(tmp.name, "t{index:05d}.nc"), want_roi=args.want_roi, want_masks=args.want_roi)
        for out in process_stream(source(), flat, 100.0, seed=7, overlap=ov, sink=sink, want_roi=args.want_roi, **kw):
            markers += sum(len(b) for b in out["beads"])
        torch.cuda.synchronize()
        return markers

    run()  # workspaces (and, with --files, the page cache: the timed pass reads what the box can keep cached)
    t0 = time.perf_counter()
    markers = run()
    dt = time.perf_counter() - t0
    tile_bytes = T * C * R * R * ty * ty * 2
    print(json.dumps({"workload": f"C5 rehearsal: {T} timepoints x {C} ch x {R}x{R} tiles of {ty}^2 (overlap {ov} -> {h}x{w}), "
                                  f"chunks of {args.chunk} timepoints from pinned host memory, num_iter={args.num_iter}",
                      "source": ("OME-BigTIFF files, one per tile position, read page by page (magnify_amd.tiff)" if args.files
                                 else "pinned host memory"), "sink": args.sink, "roi_pixels_to_sink": bool(args.want_roi),
                      "write_files_s": write_s, "streams": args.streams, "seconds": dt, "ms_per_timepoint": 1e3 * dt / T,
                      "stitched_MPs": T * C * h * w / dt / 1e6, "host_to_device_GBs": tile_bytes / dt / 1e9,
                      "markers": markers, "markers_per_s": markers / dt}))


if __name__ == "__main__":
    main()
