"""BASELINE.json configs C1-C3 through the drop-in API (mg.beads / mg.microfluidic_chip), timed on the
GPU, with the oracle's C restatement timed beside it on the host for the bead configs (SURVEY 8d:
"report MP/s and markers/s for C1-C3 in full").  C4 is bench.py's workload.

    python tests/config_table.py [--out gpurun_out/configs.json] [--num-iter 5000000] [--no-cpu]

Prints one JSON object per config; writes the list to --out.  Test infrastructure: it lives under tests/
because it times and checks against oracle/ (which only tests/, smoke() and bench.py's cpu_baseline may use).
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
sys.path.insert(0, os.path.join(ROOT, "tests"))  # synth.py

import magnify_amd as mg  # noqa: E402
from magnify_amd import utils as mg_utils  # noqa: E402
from magnify_amd.stack import synthetic_stack  # noqa: E402


def timed(fn, repeat):
    for _ in range(3):  # warm-up: workspaces, tables, the launch-sequence graphs of both output sets
        out = fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(repeat):
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return out, best


def bead_config(name, n_c, size, seed, args):
    stack, truth = synthetic_stack(1, n_c, size, size, seed=seed)
    planes = stack[0]  # (C, H, W) on the device
    host = planes.cpu().numpy()
    dims = ("channel", "y", "x") if n_c > 1 else ("y", "x")
    api_seed = 100 + seed

    def run(data):
        mg.seed(api_seed)
        return mg.beads(data=mg.DataArray(data, dims), overlap=0, num_iter=args.num_iter, search_channel=0 if n_c > 1 else None)

    xp, t_dev = timed(lambda: run(planes if n_c > 1 else planes[0]), args.repeat)
    _, t_host = timed(lambda: run(host if n_c > 1 else host[0]), args.repeat)
    m = xp.roi.sizes["mark"]
    px = n_c * size * size
    rec = {"config": name, "shape": [1, n_c, size, size], "num_iter": args.num_iter, "markers": int(m), "drawn_beads": int(len(truth)),
           "gpu_ms_device_resident": 1e3 * t_dev, "gpu_MPs": px / t_dev / 1e6, "gpu_markers_per_s": m / t_dev,
           "gpu_ms_from_host_arrays": 1e3 * t_host, "gpu_MPs_from_host_arrays": px / t_host / 1e6}
    if not args.no_cpu:
        from oracle import cport

        first = (api_seed + 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF  # magnify_amd.utils.next_seed, first draw
        t0 = time.perf_counter()
        img = cport.flatfield_correct(host[:, None, None, None], 1.0, 0.0)[:, 0, 0, 0]
        want = cport.bead_assay(img, 5, 25, 100, num_iter=args.num_iter, seed=first)
        t_cpu = time.perf_counter() - t0
        got_rc = np.column_stack([np.asarray(xp.y.values).reshape(m, -1)[:, 0], np.asarray(xp.x.values).reshape(m, -1)[:, 0]])
        same = (m == len(want["beads"]) and np.array_equal(got_rc.astype(np.int64), want["beads"][:, :2])
                and np.array_equal(np.asarray(xp.roi.values).reshape(want["roi"].shape), want["roi"])
                and np.array_equal(np.asarray(xp.fg.values).reshape(want["fg"].shape), want["fg"]))
        rec.update({"cpu_ms": 1e3 * t_cpu, "cpu_MPs": px / t_cpu / 1e6, "cpu_markers_per_s": len(want["beads"]) / t_cpu,
                    "cpu_threads": 1, "cpu_kind": "C restatement (oracle/c/ref_port.c), one assay = one thread",
                    "same_beads_roi_fg_as_gpu": bool(same)})
    return rec


def chip_config(args):
    from synth import draw_chip

    n, pitch, ty, overlap = 28, 250, 1024, 102
    canvas = draw_chip((n, n), 20, row_dist=pitch, col_dist=pitch)
    step = ty - overlap
    need = 7 * step + ty
    big = np.zeros((max(need, canvas.shape[0]), max(need, canvas.shape[1])), dtype=np.uint16)
    big[: canvas.shape[0], : canvas.shape[1]] = canvas
    tiles = np.stack([np.stack([big[r * step : r * step + ty, c * step : c * step + ty] for c in range(8)]) for r in range(8)])
    dev = torch.from_numpy(tiles).cuda()

    def run(data):
        mg.seed(3000)
        return mg.microfluidic_chip(data=mg.DataArray(data, ("row", "col", "y", "x")), shape=(n, n), overlap=overlap,
                                    row_dist=pitch, col_dist=pitch, min_button_diameter=8, max_button_diameter=30,
                                    num_iter=args.num_iter)

    xp, t_dev = timed(lambda: run(dev), max(1, args.repeat - 1))
    _, t_host = timed(lambda: run(tiles), 1)
    image = np.asarray(xp.image.values)
    side = 8 * step
    clip = overlap // 2
    x = np.asarray(xp.unstack().transpose("mark_row", "mark_col", ...).x.values).reshape(n, n)
    y = np.asarray(xp.unstack().transpose("mark_row", "mark_col", ...).y.values).reshape(n, n)
    want_x = pitch * (np.arange(n)[None, :] + 1) - clip
    want_y = pitch * (np.arange(n)[:, None] + 1) - clip
    err = float(max(np.abs(x - want_x).max(), np.abs(y - want_y).max()))
    px = 64 * ty * ty
    rec = {"config": "C3", "shape": [8, 8, ty, ty], "stitched": [side, side], "num_iter": args.num_iter, "markers": n * n,
           "max_centre_error_px": err, "gpu_ms_device_resident": 1e3 * t_dev, "gpu_MPs": px / t_dev / 1e6,
           "gpu_markers_per_s": n * n / t_dev, "gpu_ms_from_host_arrays": 1e3 * t_host,
           "gpu_MPs_from_host_arrays": px / t_host / 1e6}
    if not args.no_cpu:
        # the oracle's restatement of the chip path (stitch, flat-field, find_centers, find_rois) with its C port
        # as the circle search; same RNG streams as the GPU call (magnify_amd.utils.next_seed: 1st and 2nd draw)
        from oracle import cport
        from oracle import ref_pipeline as rp

        def c_find_circles(img_u8, low_q, high_q, grid, num_iter, min_r, max_r, min_roundness, min_dist, seed=0, **_):
            return cport.find_circles(img_u8, low_q, high_q, grid, num_iter, min_r, max_r, min_roundness, min_dist, seed=seed)

        saved = rp.find_circles
        rp.find_circles = c_find_circles
        try:
            k = 0x632BE59BD9B4E019
            s1, s2 = (3000 + k) & 0xFFFFFFFFFFFFFFFF, (3000 + 2 * k) & 0xFFFFFFFFFFFFFFFF
            tag = np.full((n, n), "default", dtype="<U200")
            t0 = time.perf_counter()
            img = rp.stitch(rp.flatfield_correct(tiles[None, None], 1.0, 0.0), overlap)[0, 0]
            min_r, max_r, chamber_r, L = rp.button_params(8, 30, 60)
            ox, oy = rp.find_centers(img[None], tag, pitch, pitch, min_r, max_r, chamber_r, 0.1, 0.9, args.num_iter, 0.2, 50, seed=s1)
            _, _, _, x_o, y_o = rp.find_rois(img[None], ox, oy, tag, [0], min_r, max_r, chamber_r, L, 0.1, args.num_iter, 0.2, seed=s2)
            t_cpu = time.perf_counter() - t0
        finally:
            rp.find_circles = saved  # other tests of the same process use the NumPy circle search
        rec.update({"cpu_ms": 1e3 * t_cpu, "cpu_MPs": px / t_cpu / 1e6, "cpu_markers_per_s": n * n / t_cpu, "cpu_threads": 1,
                    "cpu_kind": "NumPy restatement of stitch / flat-field / grid fit / refinement with the C port "
                                "(oracle/c/ref_port.c) as its circle search",
                    "same_xy_as_gpu": bool(np.allclose(x, x_o, rtol=0, atol=1e-9) and np.allclose(y, y_o, rtol=0, atol=1e-9)),
                    "same_image_as_oracle_stitch": bool(image.shape == img.shape and np.array_equal(image, img))})
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/configs.json")
    ap.add_argument("--num-iter", type=int, default=5_000_000)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--only", default="C1,C2,C3")
    args = ap.parse_args()
    from magnify_amd import hotpath

    hotpath.require_gpu()
    recs = []
    for name in args.only.split(","):
        if name == "C1":
            recs.append(bead_config("C1", 1, 2048, 1000, args))
        elif name == "C2":
            recs.append(bead_config("C2", 4, 4096, 2000, args))
        elif name == "C3":
            recs.append(chip_config(args))
        print(json.dumps(recs[-1]), flush=True)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(recs, f, indent=1)
    _ = mg_utils


if __name__ == "__main__":
    main()
