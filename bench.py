#!/usr/bin/env python3
"""bench.py -- throughput of the marker-detection hot path on MI355X.

One "step" = one pass of flat-field + stitch -> bead detection -> fg/bg segmentation -> per-ROI
reduction over one synthetic (T x C x H x W) uint16 stack that is already resident in HBM.
Workload at N=1: BASELINE.json's 64 x 4 x 4096 x 4096 stack (C4), per-timepoint detection (mode P),
search channel 0, the reference's default 5 000 000 RANSAC iterations per searched plane.
N > 1, one rank per GPU -- started by torch.distributed.run, or by this script itself: a parent
that never touches the GPU spawns the N ranks (magnify_amd/launch.py) and relays rank 0's line.
  --scaling weak (default): every rank owns its own 64-timepoint block, runs the chain locally and
      all-gathers the final marker table over RCCL; for N > 1 the line also carries a `strong` object,
      measured right after the timed region.
  --scaling strong: north_star's C4 -- ONE 64-timepoint stack, contiguous shards of 64/N timepoints
      per rank (distributed.shard_range), same all-gather.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def algorithmic_bytes(stage, p):
    """Algorithmic HBM bytes of a stage PER STEP, summed over its launches (DESIGN.md 'Kernels');
    p = workload numbers of the step.  A launch's share is this divided by the launches per step."""
    n, planes, c = p["h"] * p["w"], p["search_planes"], p["n_c"]
    table = {
        "mg_flatfield_max": 2 * c * n * p["n_t"],            # read u16, whole stack
        "mg_flatfield_apply_stitch": 4 * c * n * p["n_t"],   # read u16 + write u16
        "mg_to_uint8_blur": 3 * planes * n,                 # read u16, write blurred u8
        "mg_to_uint8_blur_hist": 3 * planes * n,            # the same bytes: the histogram is taken from registers
        "mg_scharr_hist": 1 * planes * n * (p["hist_passes"] - 1),  # window passes only: read u8
        "mg_canny_nms": (1 + 5 / 8) * planes * n,           # read u8, write weak + strong bitmaps + 3 orientation bit planes
        "mg_canny_hysteresis": (3 / 8) * planes * n * p["sweeps"],  # weak + strong bits in, strong out, per sweep
        "mg_edge_angles": p["edges"] * (8 + 9 + 4),          # coordinate, 3x3 blurred neighbourhood, angle
        "mg_edge_grid": 2 * planes * n / 8 + 8 * p["edges"],  # bitmap twice (count, fill), write coords
        "mg_candidate_circles": 28 * planes * p["num_iter"],  # 3 coordinate reads (8 B) + the 32-bit key
        # keys read once, the unique keys written once (tile by tile, from LDS)
        "mg_bitmap_to_circles": 4 * planes * p["num_iter"] + 4 * p["unique"],
        # keyed scoring: every unique key once + the four bit planes (edges + 3 orientation planes) of the searched
        # planes once + the survivors' records and scores; its real bound is the LDS (one byte read per perimeter
        # point), the HBM bytes are small
        "mg_score_circles": p["unique"] * 4 + planes * n / 8 * 4 + p.get("scored", 0) * 16,
        "mg_nms_rounds": p["alive"] * p["ring_len"] * 16 * p["nms_rounds"],
        "mg_collect_circles": p["alive"] * 4 + p["markers"] * 16,
        "mg_circle_labels": p["markers"] * p["mean_disk"] * 8,
        "mg_roi_gather_reduce_batched": p["markers"] * p["L"] ** 2 * (4 * c + 6),
        "mg_roi_segment_reduce": p["markers"] * p["L"] ** 2 * (4 * c + 2),  # no label map: pixels in/out + masks out
        "mg_roi_window_order": p["markers"] * (12 + 4),       # bead triples in, order out
        "mg_plane_minmax": 2 * planes * n,
    }
    return table.get(stage)


def cpu_baseline(args, stack, flat_np, seeds, gpu_counts, gpu_fg_sums):
    """The oracle's C restatement of the reference path (oracle/c/ref_port.c, kind 'port') timed on
    this host on a bounded sample of the SAME workload: the first timepoints of the bench stack (one
    assay per OpenMP thread, at most 16 threads = the box's CPU share), same flat-field, same
    RANSAC budget and the seeds of the last GPU step, so the bead counts and the fg-sum checksums
    of the two paths can be compared as well."""
    import numpy as np

    from oracle import cport

    cores = max(1, min(16, len(os.sched_getaffinity(0)), cport.max_threads(), args.timepoints))
    n = min(args.timepoints, cores * max(1, args.cpu_assays_per_core))
    sample = stack[:n].cpu().numpy()
    t0 = time.perf_counter()
    total, counts, sums = cport.run_stack(sample, flat_np, 100.0, 5, 25, 100, seeds[:n], num_iter=args.num_iter,
                                          n_threads=cores)
    dt = time.perf_counter() - t0
    mp = n * args.channels * args.size * args.size / 1e6
    agree = bool(np.array_equal(counts, np.asarray(gpu_counts[:n])) and np.array_equal(sums, np.asarray(gpu_fg_sums[:n])))
    return {"value": mp / dt, "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"first {n} of the step's {args.timepoints} timepoints ({args.channels} ch x {args.size}x{args.size} "
                      f"uint16 each, num_iter={args.num_iter}), one assay per OpenMP thread on {cores} threads, "
                      f"{total} markers, {dt:.1f} s (C restatement oracle/c/ref_port.c)",
            "markers_per_s": total / dt, "same_markers_and_sums_as_gpu": agree}


# what actually bounds a stage when it is not HBM bandwidth; `roofline` still prices it
# against the HBM peak, as the contract asks
STAGE_NOTES = {
    "mg_score_circles": "LDS-bound: one random LDS byte read per perimeter point and circle (~8 LDS cycles per wave "
                        "read measured with 32 random dwords per half-wave, tools/micro/walk_bench.hip); its HBM bytes "
                        "are the 4-byte circle keys and the bit planes only",
    "mg_canny_nms": "VALU-issue bound (OpenCV's sector compares / selects)",
    "mg_candidate_circles": "VALU bound (three float64 divisions per RANSAC iteration)",
}

# stages launched outside the finder's chain: timed live in the timed region even when the chain is a graph replay
LIVE_STAGES = ["mg_flatfield_max", "mg_flatfield_apply_stitch", "mg_roi_segment_reduce", "mg_counts_to_offsets",
               "mg_roi_window_order"]

STREAM_STAGES = ["mg_flatfield_max", "mg_flatfield_apply_stitch", "mg_to_uint8_blur", "mg_to_uint8_blur_hist", "mg_scharr_hist", "mg_canny_nms",
                 "mg_canny_hysteresis", "mg_edge_grid", "mg_edge_angles", "mg_circle_labels",
                 "mg_roi_gather_reduce_batched", "mg_roi_window_order", "mg_roi_segment_reduce"]


def source_hash():
    """Hash of the sources that decide what the kernels do and how they are launched: the PMC traffic file under
    profiles/ carries the hash it was measured on, and `traffic` is only reported while it still matches
    (the GPU box has no .git to ask for a commit)."""
    import glob
    import hashlib

    hsh = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "magnify_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "magnify_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h"))
                   + [os.path.join(ROOT, "magnify_amd", f) for f in ("hotpath.py", "stack.py")])
    for f in files:
        hsh.update(os.path.basename(f).encode())
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def load_traffic(shape_key):
    """Newest profiles/r*_pmc_traffic.json measured on exactly these sources and this workload, or None."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            rec = json.load(open(f))
        except (OSError, ValueError):
            continue
        if rec.get("source_hash") == source_hash() and tuple(rec.get("shape", ())) == tuple(shape_key):
            return rec["stages"], os.path.basename(f)
    return None, None


def stage_report(stages, steps, p, pmc, stream_bytes):
    """Per-stage table (time, algorithmic bytes, achieved GB/s, PMC traffic), the `roofline` object of
    the stage with the largest total time, and the streaming part (everything that is not RANSAC
    scoring / suppression) against SURVEY 8d's byte count."""
    total_ms = sum(v[0] for v in stages.values())
    breakdown = {}
    for k, v in sorted(stages.items(), key=lambda kv: -kv[1][0]):
        ab = algorithmic_bytes(k, p)
        ms_step = v[0] / steps
        breakdown[k] = {"ms_per_step": round(ms_step, 3), "launches_per_step": v[1] / steps,
                        "ms_avg_launch": round(v[0] / v[1], 4),
                        "algorithmic_GB_per_step": round(ab / 1e9, 3) if ab else None,
                        "achieved_GBs": round(ab / (ms_step / 1e3) / 1e9, 1) if ab and ms_step else None,
                        "hbm_traffic_GB_per_step": round(pmc[k]["hbm_bytes_per_step"] / 1e9, 3) if pmc and k in pmc else None}
    dom, (dom_ms, dom_n) = max(stages.items(), key=lambda kv: kv[1][0])
    ab_step = algorithmic_bytes(dom, p)                  # bytes per step, all launches of the stage
    launches_per_step = dom_n / steps
    ab = ab_step / launches_per_step if ab_step else None  # bytes of one launch
    avg_s = dom_ms / dom_n / 1e3                          # average duration of one launch
    achieved = ab / avg_s / 1e9 if ab else None
    traffic = pmc[dom]["hbm_bytes_per_step"] / launches_per_step if pmc and dom in pmc else None
    roofline = {"kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                "algorithmic_bytes_per_launch": ab, "avg_launch_ms": dom_ms / dom_n,
                "launches_per_step": launches_per_step, "share_of_kernel_time": dom_ms / total_ms,
                "note": STAGE_NOTES.get(dom),
                # what the counters show to limit the stage (profiles/): `bound` keeps the contract's vocabulary
                "limiter": {"mg_score_circles": "lds", "mg_canny_nms": "valu", "mg_candidate_circles": "valu"}.get(dom, "hbm")}
    stream_ms = sum(stages[s][0] for s in STREAM_STAGES if s in stages) / steps
    streaming = {"ms_per_step": stream_ms, "algorithmic_bytes": stream_bytes,
                 "achieved_GBs": stream_bytes / (stream_ms / 1e3) / 1e9 if stream_ms else None,
                 "frac_of_peak": stream_bytes / (stream_ms / 1e3) / 1e9 / HBM_PEAK_GBS if stream_ms else None}
    return breakdown, roofline, streaming, total_ms


def hbm_copy_ceiling(dev, n_bytes=1 << 30, reps=10):
    """Measured streaming ceiling of this box, with the library's own streaming kernel (mg_stream_probe:
    16 bytes per lane): a read + write copy of 1 GiB (bytes read + bytes written), a read-only and a write-only pass,
    HIP-event time of ``reps`` launches each, in two launch shapes -- a resident round of workgroups with four loads in
    flight per lane (how the hot path's streaming passes are launched) and one access per lane with as many workgroups
    as that takes.  The ceiling reported is the faster shape's.  Outside the timed region."""
    import torch

    from magnify_amd import _native as nat
    from magnify_amd import hotpath as hp

    try:
        src = torch.empty(n_bytes, dtype=torch.uint8, device=dev).fill_(1)
        dst = torch.empty_like(src)
    except RuntimeError:
        return None
    sink = torch.zeros(4 * (n_bytes // 16 // 256 + 1), dtype=torch.int32, device=dev)
    out = {}
    for shape, blocks in (("resident_round", 256 * 8), ("one_trip", 0)):
        for name, mode, moved in (("copy", 0, 2 * n_bytes), ("read", 1, n_bytes), ("write", 2, n_bytes)):
            call = lambda: nat.check(nat.lib().mg_stream_probe(src.data_ptr(), dst.data_ptr(), n_bytes, mode, sink.data_ptr(),  # noqa: E731
                                                              blocks, hp._stream()), "mg_stream_probe")
            call()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                call()
            e1.record()
            torch.cuda.synchronize()
            out.setdefault(shape, {})[name + "_GBs"] = moved / (e0.elapsed_time(e1) / reps) / 1e6
    for name in ("copy", "read", "write"):
        out[name + "_GBs"] = max(out[s][name + "_GBs"] for s in ("resident_round", "one_trip"))
    out["GBs"] = out["copy_GBs"]
    out["what"] = (f"mg_stream_probe over {n_bytes >> 20} MiB: copy = bytes read + written; the library's own 16-byte-per-lane "
                   "streaming kernel, the faster of two launch shapes")
    return out


def per_assay_fg_sums(out, n_assays):
    """Sum of the foreground sums of every assay's markers (exact integers held in float64)."""
    import numpy as np

    fg = out["sums"][..., 0].sum(dim=(1, 2)).cpu().numpy()
    off = out["offsets"]
    return [int(fg[off[a]:off[a + 1]].sum()) for a in range(n_assays)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--timepoints", type=int, default=64, help="timepoints per GPU")
    ap.add_argument("--channels", type=int, default=4)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--num-iter", type=int, default=5_000_000)
    ap.add_argument("--plane-batch", type=int, default=0, help="searched planes per kernel batch (0 = all)")
    ap.add_argument("--streams", type=int, default=1,
                    help="sub-batches of assays on separate HIP streams / host threads (1 = one batch on one stream: "
                         "fastest since the kernels were tightened; 4 was +7 % before that)")
    ap.add_argument("--sub-batches", type=int, default=0, help="sub-batches of assays (0 = one per stream)")
    ap.add_argument("--cpu-assays-per-core", type=int, default=2)
    ap.add_argument("--from-host", action="store_true",
                    help="the stack starts every step in pinned HOST memory (PCIe-inclusive rate; not the headline)")
    ap.add_argument("--noiseless", action="store_true",
                    help="the reference tests' kind of image: zero background, beads of value 1000 (not the headline)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-isolated", action="store_true", help="skip the extra single-stream pass")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --timepoints per GPU; strong: --timepoints in total, sharded over the GPUs")
    ap.add_argument("--no-strong-extra", action="store_true",
                    help="weak scaling with N > 1: skip the extra strong-scaling measurement")
    args = ap.parse_args()

    from magnify_amd import launch

    if args.gpus > 1 and not launch.launched_by_torchrun():
        # this process has not touched the GPU (and never will): it starts one child per rank
        raise SystemExit(launch.spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus))

    import numpy as np
    import torch

    from magnify_amd import distributed as mgd
    from magnify_amd import hotpath as hp
    from magnify_amd.stack import StackProcessor, synthetic_stack
    from synth import vignette

    rank, world, local = mgd.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    hp.require_gpu()
    dev = torch.device("cuda", local)
    C, S = args.channels, args.size
    strong = args.scaling == "strong"
    if strong:   # one stack of --timepoints, this rank owns [lo, hi)
        lo, hi = mgd.shard_range(args.timepoints, rank, world)
    else:        # every rank owns its own block of --timepoints
        lo, hi = rank * args.timepoints, (rank + 1) * args.timepoints
    T = hi - lo
    if T == 0:
        raise SystemExit(f"rank {rank} owns no timepoint: --timepoints {args.timepoints} < --gpus {world}")

    # the stack is the concatenation of the ranks' blocks; a block's content only depends on where it starts
    # (N = 1: seed 4000, the whole stack, in both modes)
    stack, truth = synthetic_stack(T, C, S, S, seed=4000 + (lo if strong else 100 * rank), device=dev,
                                   noiseless=args.noiseless)
    flat_np = vignette((S, S))
    flat = torch.from_numpy(flat_np).to(dev)

    def make_proc(n_t):
        return StackProcessor(n_t, C, S, S, num_iter=args.num_iter, min_bead_diameter=10, max_bead_diameter=50,
                              search_channels=(0,), mode="P", plane_batch=args.plane_batch or None, device=dev,
                              n_streams=args.streams, sub_batches=args.sub_batches or None)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(dt):
        if world > 1:
            gloo = torch.distributed.get_backend() == "gloo"
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if gloo else dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def run_steps(proc, src, first_assay, with_timer):
        """W untimed + K timed steps; a step = the chain on this rank's block + the marker-table all-gather.
        The detection RNG stream of an assay depends on its GLOBAL index only (seed + 1000003 * index)."""
        def step(seed):
            out = proc(src, flat, 100.0, seed=(seed + 1000003 * first_assay) & 0xFFFFFFFFFFFFFFFF)
            table = mgd.gather_marker_table(mgd.marker_table(out, first_assay, C, dev))
            return out, table

        for i in range(args.warmup):
            out, table = step(i)
        # live HIP-event timing of everything launched around the finder's optimistic chain (flat-field passes, ROI
        # pass); the chain itself may run as one hipGraph launch, which has no room for events between its kernels
        timer = hp.StageTimer(allow_graphs=True) if with_timer else None
        barrier()
        hp.set_timer(timer)
        t0 = time.perf_counter()
        for i in range(args.steps):
            out, table = step(args.warmup + i)
        barrier()
        dt = time.perf_counter() - t0
        hp.set_timer(None)
        live = timer.summary() if with_timer else None
        stages = None
        if with_timer:
            # per-stage table: the same steps once more with an event pair around EVERY C-ABI call (eager launches,
            # no graph), outside the timed region; the stages timed live above keep their live numbers
            n_prof = max(2, min(args.steps, 5))
            full = hp.StageTimer()
            torch.cuda.synchronize()
            hp.set_timer(full)
            for i in range(n_prof):
                step(args.warmup + args.steps - 1)
            torch.cuda.synchronize()
            hp.set_timer(None)
            stages = {k: (v[0] * args.steps / n_prof, v[1] * args.steps / n_prof) for k, v in full.summary().items()}
            for k in LIVE_STAGES:
                if k in live:
                    stages[k] = live[k]
            out, table = step(args.warmup + args.steps - 1)  # (the results reported are those of the last timed step's seed)
        return out, table, max_over_ranks(dt), stages

    proc = make_proc(T)
    src = stack.cpu().pin_memory() if args.from_host else stack
    out, table, dt, stages = run_steps(proc, src, lo, True)
    n_total = args.timepoints if strong else world * args.timepoints  # timepoints all ranks processed per step

    strong_extra = None
    if world > 1 and not strong and not args.no_strong_extra and args.timepoints >= world:
        # the same ranks once more on north_star's C4 proper: ONE --timepoints stack, 1/N of it per rank
        # (the first timepoints of this rank's block stand in for its shard)
        slo, shi = mgd.shard_range(args.timepoints, rank, world)
        proc_s = make_proc(shi - slo)
        _, table_s, dt_s, _ = run_steps(proc_s, src[: shi - slo], slo, False)
        strong_extra = {"value": args.timepoints * C * S * S / 1e6 / (dt_s / args.steps), "unit": "MP/s",
                        "ms_per_step": dt_s / args.steps * 1e3, "timepoints_total": args.timepoints,
                        "timepoints_per_gpu": shi - slo, "markers": int(table_s.shape[0]),
                        "what": "strong scaling: one stack of --timepoints sharded over the ranks (same steps / warmup)"}
        del proc_s

    markers_local = int(sum(len(b) for b in out["beads"]))
    markers_total = int(table.shape[0])
    mp_total = n_total * C * S * S / 1e6
    ms_per_step = dt / args.steps * 1e3

    if rank == 0:
        # workload numbers for the algorithmic-byte table (last step, rank 0)
        f = proc.finder
        if proc.n_streams > 1:  # one record per (sub-batch, search channel) of the last step
            dev_counts = torch.stack([x[0] for x in proc.step_stats]).sum(dim=0).tolist()
            unique, alive, scored = (int(v) for v in dev_counts)
            edges = sum(x[1] for x in proc.step_stats)
            fstats = [x[2] for x in proc.step_stats]
            search_planes = proc.n_assays * len(proc.search_channels)
        else:
            unique, alive, scored = int(f.num_circles.sum().item()), int(f.num_alive.sum().item()), int(f.num_scored.sum().item())
            edges, fstats, search_planes = int(f.n_edges_host.sum()), [f.stats], f.P
        finders = list(getattr(proc, "finders", None) or [f])
        chain = {k: sum(x.calls[k] for x in finders) for k in ("optimistic", "repaired", "checked", "followed_again")}
        per_starts = f.per_starts.cpu().numpy()
        mean_perimeter = float(np.mean(np.diff(per_starts)))
        p = {"h": S, "w": S, "n_c": C, "n_t": T, "search_planes": search_planes, "num_iter": args.num_iter,
             "hist_passes": max(x.get("hist_passes", 1) for x in fstats),
             "sweeps": max(x.get("hysteresis_sweeps", 1) for x in fstats),
             "nms_rounds": max(x.get("nms_rounds", 1) for x in fstats),
             "edges": edges, "bitmap_words": f.bitmap_words, "unique": unique, "alive": alive, "scored": scored,
             "mean_perimeter": mean_perimeter, "ring_len": len(hp.nat.circle_points(proc.min_r, True)),
             "markers": markers_local, "mean_disk": 600, "L": proc.L}
        # HBM bytes per step from the committed rocprofv3 PMC passes: only when they were taken on these sources
        pmc, pmc_file = load_traffic((T, C, S, args.num_iter))
        n_all, n_s = T * C * S * S, search_planes * S * S
        # SURVEY.md 8d's count minus the label map this build no longer writes (4 B/px of the searched
        # planes) or reads (4 B per window pixel): masks come straight from the bead tables
        # ... and minus the histogram pass's read of the blurred image (1 B/px) where the one-pass blur + histogram
        # kernel ran (mg_to_uint8_blur_hist takes the magnitudes from registers): bytes saved are not bytes moved
        one_pass_hist = "mg_to_uint8_blur_hist" in stages and "mg_scharr_hist" not in stages
        stream_bytes = 6 * n_all + (7 if one_pass_hist else 8) * n_s + markers_local * proc.L**2 * (4 * C + 2)
        breakdown, roofline, streaming, total_ms = stage_report(stages, args.steps, p, pmc, stream_bytes)
        # the same stages against SURVEY 8d's own count, which includes the label map (4 B written per searched pixel,
        # 4 B read per window pixel) that this build never materialises: bytes saved, not bytes moved
        survey_stream = 6 * n_all + 12 * n_s + markers_local * proc.L**2 * (4 * C + 6)
        if streaming.get("ms_per_step"):
            streaming["survey_8d_bytes_incl_label_map"] = survey_stream
            streaming["frac_of_peak_by_survey_bytes"] = survey_stream / (streaming["ms_per_step"] / 1e3) / 1e9 / HBM_PEAK_GBS
        roofline["traffic_source"] = pmc_file  # None: no PMC pass on these exact sources under profiles/
        # SURVEY 8d's byte count of the whole step, B = 6 N_all + 12 N_s + K_u (12 + 5 P) + M L^2 (4 C + 6), with K_u
        # the circles that reach the exact sum; against the wall-clock step (kernels + host round trips)
        b8d = 6 * n_all + 12 * n_s + scored * (12 + 5 * mean_perimeter) + markers_local * proc.L**2 * (4 * C + 6)
        whole = {"algorithmic_bytes_survey_8d": b8d, "ms_per_step": ms_per_step,
                 "achieved_GBs": b8d / (ms_per_step / 1e3) / 1e9, "frac_of_peak": b8d / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS}
        isolated = None
        if world == 1 and proc.n_streams > 1 and not args.no_isolated:
            # the same step once more on ONE stream (untimed for `value`): kernel durations without the
            # overlap of the four detection streams, for the per-stage roofline table
            proc1 = StackProcessor(T, C, S, S, num_iter=args.num_iter, min_bead_diameter=10, max_bead_diameter=50,
                                   search_channels=(0,), mode="P", device=dev, n_streams=1)
            proc1(stack, flat, 100.0, seed=0)
            t1 = hp.StageTimer()
            torch.cuda.synchronize()
            hp.set_timer(t1)
            w0 = time.perf_counter()
            for i in range(2):
                proc1(stack, flat, 100.0, seed=args.warmup + args.steps - 1)
            torch.cuda.synchronize()
            w1 = time.perf_counter() - w0
            hp.set_timer(None)
            bd1, rf1, st1, tot1 = stage_report(t1.summary(), 2, p, pmc, stream_bytes)
            isolated = {"note": "one HIP stream, no overlap between kernels; not the timed region",
                        "ms_per_step": w1 / 2 * 1e3, "kernel_ms_per_step": tot1 / 2, "roofline": rf1,
                        "streaming_part": st1, "stages": bd1}
            del proc1
        result = {
            "metric": "megapixels/sec through flatfield+segment+ROI-reduce; markers/sec",
            "value": mp_total / (dt / args.steps), "unit": "MP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "u16",
            "data": "synthetic" + (" noiseless" if args.noiseless else "") + (" (host-resident, PCIe-inclusive)" if args.from_host else ""),
            "config": {"workload": (f"C4: {args.timepoints} timepoints x {C} ch x {S}x{S} uint16 "
                                    + ("in total" if strong else "per GPU") + ", mode P (per-timepoint "
                                    f"detection), search channel 0, num_iter={args.num_iter}, vignette flat-field, "
                                    f"dark=100, roi_length={proc.L}"),
                       "timepoints_per_gpu": T, "timepoints_total": n_total, "parallelism": f"time-shard x{world}"},
            "ranks": {"world_size": torch.distributed.get_world_size() if world > 1 else 1,
                      "backend": (("rccl" if torch.distributed.get_backend() == "nccl" else torch.distributed.get_backend())
                                  if world > 1 else None),
                      "shared_gpu": os.environ.get("MG_SHARE_GPU") == "1",
                      "launcher": "self" if os.environ.get("MG_LAUNCHED") == "1" else ("torchrun" if world > 1 else None)},
            "strong": strong_extra,
            "markers_per_s": markers_total / (dt / args.steps), "markers": markers_total,
            "roofline": roofline,
            "whole_step": whole,
            "streaming_part": streaming,
            "stages": breakdown,
            "isolated": isolated,
            "stats": {"unique_circles": unique, "scored_exactly": scored,
                      "streams": proc.n_streams, "sub_batches": len(getattr(proc, "ranges", [0])),
                      "alive_circles": alive, "edges": p["edges"],
                      "hysteresis_sweeps": p["sweeps"], "nms_rounds": p["nms_rounds"],
                      # calls of find() since start-up (warm-up included): with ONE host round trip (optimistic),
                      # with a repair after it, with the three round trips of the checked chain; followed_again:
                      # optimistic calls whose ROI pass had to run a second time (output regathered)
                      "find_calls": chain,
                      # optimistic calls whose ~40 launches went out as one hipGraph replay
                      "graph_replays": sum(getattr(x, "graph_replays", 0) for x in finders),
                      # the processor's first call timed its passes into several image blocks / ROI output sets and
                      # kept the fastest (stack.StackProcessor placement trial; None: no trial)
                      "placement": getattr(proc, "placement", None),
                      "stage_timing": "HIP events on the launch stream: live in the timed region for " + ", ".join(LIVE_STAGES)
                                      + "; the stages inside the finder's chain from the same steps re-run eagerly right after it",
                      "kernel_ms_per_step": total_ms / args.steps},
        }
        result["hbm_copy_ceiling"] = hbm_copy_ceiling(dev)
        for part in (roofline, streaming, whole):
            if part and result["hbm_copy_ceiling"]:
                gbs = part.get("achieved", part.get("achieved_GBs"))
                part["frac_of_measured_copy"] = gbs / result["hbm_copy_ceiling"]["GBs"] if gbs else None
        if not args.no_cpu and world == 1:
            last = args.warmup + args.steps - 1  # seed of the step whose results are in `out`
            seeds = [(last + 1000003 * a) & 0xFFFFFFFFFFFFFFFF for a in range(T)]
            result["cpu_baseline"] = cpu_baseline(args, stack, flat_np, seeds, [len(b) for b in out["beads"]],
                                                  per_assay_fg_sums(out, T))
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
