"""Config C5 measured: a TILED time series (T x C x rows x cols tiles of tile^2 with overlap) arrives chunk by chunk
-- from pinned host buffers, or (--files DIR) from OME-BigTIFF files, one per tile position with (time, channel) pages,
read through magnify_amd.tiff / reader.iter_time_chunks --; `stack.process_stream` uploads a chunk, crops / joins /
corrects the tiles on the device and runs the hot path; the stitched assay never exists on the host.
--sink host|save keeps / writes every timepoint's results (magnify_amd.sink); --want-roi sends the ROI pixel stacks and
masks there too (the reference caches `roi` for every assay, find.py:589-604).  PCIe- (and file-) inclusive throughput
in stitched megapixels per second (the figure that is NOT bench.py's `value`).

--gpus N: ONE series, its time axis split over N ranks (SURVEY 8e, C5's partition): a parent that never touches the
GPU starts the ranks (magnify_amd.launch), rank 0 writes the files, every rank streams its block
(`distributed.stream_series`), the run ends with the marker-table all-gather; the line reports the slowest rank.
On a one-GPU box the ranks share cuda:0 over gloo (a rehearsal, the line says so).

    python tools/c5_stream_bench.py [--timepoints 64] [--chunk 8] [--grid 4] [--tile 1126] [--overlap 102] [--files DIR]
                                    [--sink save --want-roi] [--gpus N] [--reader-only]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse():
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
    ap.add_argument("--workers", type=int, default=None, help="reader threads per rank")
    ap.add_argument("--gpus", type=int, default=1, help="ranks the time axis is split over")
    ap.add_argument("--reader-only", action="store_true", help="also time the reader alone (no GPU work): where a file-bound run loses its time")
    ap.add_argument("--save-dir", default=None, help="where --sink save writes (default: a temporary directory)")
    ap.add_argument("--writers", type=int, default=8, help="files a SaveSink writes side by side")
    return ap.parse_args()


def main():
    args = parse()
    from magnify_amd import launch

    if args.gpus > 1 and not launch.launched_by_torchrun():
        # the parent never initialises HIP: it only starts the ranks and relays rank 0's line
        raise SystemExit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=1500))

    import numpy as np
    import torch

    from magnify_amd import distributed as mgd
    from magnify_amd.stack import process_stream, stitched_shape, synthetic_stack

    rank, world, _ = mgd.init_from_env()
    T, C, R, ty, ov = args.timepoints, args.channels, args.grid, args.tile, args.overlap
    h, w = stitched_shape(R, R, ty, ty, ov)
    clip, rem = ov // 2, ov % 2
    step = ty - 2 * clip - rem
    side = (R - 1) * step + ty  # a canvas large enough to cut overlapping tiles from: tile (r, c) starts at (r step, c step)
    lo, hi = mgd.shard_range(T, rank, world)
    if world > 1 and (lo % args.chunk or (hi - lo) % args.chunk):
        raise SystemExit("--gpus: timepoints / ranks must be a multiple of --chunk (the chunks' content depends on their start)")

    def make_chunk(t0):
        n = min(args.chunk, T - t0)
        canvas, _ = synthetic_stack(n, C, side, side, seed=5000 + t0)
        tiles = torch.empty((n, C, R, R, ty, ty), dtype=torch.uint16, device="cuda")
        for r in range(R):
            for c in range(R):
                tiles[:, :, r, c] = canvas[:, :, r * step:r * step + ty, c * step:c * step + ty]
        return tiles.cpu()

    write_s = None
    chunks = None
    if args.files:
        if rank == 0:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from tiffwrite import ome_xml, write_tiff

            os.makedirs(args.files, exist_ok=True)
            whole = [make_chunk(t0) for t0 in range(0, T, args.chunk)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for r in range(R):
                for c in range(R):
                    pages = [whole[t // args.chunk][t % args.chunk, ch, r, c].numpy() for t in range(T) for ch in range(C)]
                    write_tiff(os.path.join(args.files, f"acq_r{r}_c{c}.ome.tif"), pages, bigtiff=True,
                               description=ome_xml(size_c=C, size_t=T, size_y=ty, size_x=ty))
            write_s = time.perf_counter() - t0
            del whole
        if world > 1:
            torch.distributed.barrier()
    else:
        chunks = [make_chunk(t0).pin_memory() for t0 in range(lo, hi, args.chunk)]  # this rank's block of the series
    torch.cuda.synchronize()
    yy, xx = np.mgrid[0:ty, 0:ty]
    flat = (1 - 0.15 * (((yy - (ty - 1) / 2) / (ty / 2)) ** 2 + ((xx - (ty - 1) / 2) / (ty / 2)) ** 2)).astype(np.float32)
    kw = dict(num_iter=args.num_iter, search_channels=(0,), n_streams=args.streams)
    pattern = os.path.join(args.files, "acq_r(row)_c(col).ome.tif") if args.files else None
    workers = args.workers if args.workers else max(2, min(16, (os.cpu_count() or 1) // world))

    import tempfile

    import magnify_amd as mg

    tmp = None
    save_dir = args.save_dir
    if args.sink == "save" and save_dir is None:
        tmp = tempfile.TemporaryDirectory(prefix="c5_results_")
        save_dir = tmp.name
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)

    def make_sink():
        if args.sink == "host":
            return mg.HostSink(want_roi=args.want_roi, want_masks=args.want_roi)
        if args.sink == "save":
            return mg.SaveSink(os.path.join(save_dir, "t{index:05d}.nc"), want_roi=args.want_roi, want_masks=args.want_roi,
                               writers=args.writers)
        return None

    sink_stats = {}

    def run():
        sink = make_sink()
        try:
            return stream(sink)
        finally:
            if sink is not None:
                sink_stats.clear()
                sink_stats.update({k: round(v, 4) if isinstance(v, float) else v for k, v in sink.stats.items()})

    def stream(sink):
        if pattern:
            table, _ = mgd.stream_series(pattern, args.chunk, flat, 100.0, seed=7, sink=sink, rank=rank, world=world,
                                         workers=workers, overlap=ov, want_roi=args.want_roi, **kw)
            markers = int(table.shape[0])
        else:
            tables = []
            for out in process_stream(iter(chunks), flat, 100.0, seed=7, overlap=ov, sink=sink, want_roi=args.want_roi,
                                      first_timepoint=lo, **kw):
                tables.append(mgd.marker_table(out, out["first_timepoint"], C, torch.device("cuda")))
            markers = int(mgd.gather_marker_table(torch.cat(tables)).shape[0])
        torch.cuda.synchronize()
        return markers

    def timed(fn):
        """fn's wall time, the slowest rank's (barrier before, max all-reduce after: a host tensor over gloo, a device
        tensor over RCCL)."""
        if world > 1:
            torch.distributed.barrier()
        t0 = time.perf_counter()
        res = fn()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        if world > 1:
            dt = dt if torch.distributed.get_backend() == "gloo" else dt.cuda()
            torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
        return res, float(dt.item())

    run()  # workspaces (and, with --files, the page cache: the timed pass reads what the box can keep cached)
    markers, dt = timed(run)
    reader_s = None
    if args.reader_only and pattern:
        from magnify_amd import reader

        def read_all():
            n = 0
            for _, _, block in reader.iter_time_chunks(pattern, args.chunk, pinned=True, workers=workers, rank=rank, world=world):
                n += block.numel()
            return n

        read_all()
        _, reader_s = timed(read_all)
    tile_bytes = T * C * R * R * ty * ty * 2
    if rank == 0:
        print(json.dumps({
            "workload": f"C5 rehearsal: {T} timepoints x {C} ch x {R}x{R} tiles of {ty}^2 (overlap {ov} -> {h}x{w}), "
                        f"chunks of {args.chunk} timepoints, num_iter={args.num_iter}",
            "source": ("OME-BigTIFF files, one per tile position, (time, channel) pages read as byte runs by "
                       f"{workers} native threads per rank (mg_host_read_runs)" if args.files else "pinned host memory"),
            "sink": args.sink, "roi_pixels_to_sink": bool(args.want_roi), "write_files_s": write_s, "streams": args.streams,
            "ranks": {"world_size": world, "backend": torch.distributed.get_backend() if world > 1 else None,
                      "shared_gpu": os.environ.get("MG_SHARE_GPU") == "1", "timepoints_per_rank": hi - lo},
            "seconds": dt, "ms_per_timepoint": 1e3 * dt / T, "stitched_MPs": T * C * h * w / dt / 1e6,
            "host_to_device_GBs": tile_bytes / dt / 1e9, "markers": markers, "markers_per_s": markers / dt,
            "sink_writer": sink_stats or None,
            "reader_alone": None if reader_s is None else {"seconds": reader_s, "ms_per_timepoint": 1e3 * reader_s / T,
                                                           "GBs": tile_bytes / reader_s / 1e9, "workers_per_rank": workers}}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
