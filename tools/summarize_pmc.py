"""FETCH_SIZE / WRITE_SIZE passes of rocprofv3 -> HBM bytes per bench step and stage (JSON on stdout).
usage: python tools/summarize_pmc.py <tag> <pmc_fetch_dir> <pmc_write_dir> <steps the profiled process ran in all>
FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 bytes); both counters
are in KiB.  The file carries the hash of the kernel / launch sources it was measured on (bench.source_hash)."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

tag, fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
STAGE_OF = {"k_flatfield_max": "mg_flatfield_max", "k_flat_rcmax": "mg_flatfield_max", "k_apply_stitch": "mg_flatfield_apply_stitch",
            "k_u8_blur": "mg_to_uint8_blur", "k_blur_hist": "mg_to_uint8_blur_hist", "k_scharr_hist": "mg_scharr_hist", "k_hist_reduce": "mg_to_uint8_blur_hist",
            "k_edge_thresholds": "mg_edge_thresholds", "k_window_resolve": "mg_edge_thresholds", "k_canny_nms": "mg_canny_nms",
            "k_hysteresis": "mg_canny_hysteresis", "k_cell_": "mg_edge_grid", "k_edge_angles": "mg_edge_angles",
            "k_candidates": "mg_candidate_circles", "k_layer_": "mg_bitmap_to_circles", "k_tile_": "mg_bitmap_to_circles",
            "k_score_tiles": "mg_score_circles", "k_prefilter": "mg_score_circles", "k_exact": "mg_score_circles",
            "k_nms": "mg_nms_rounds", "k_collect": "mg_collect_circles", "k_clamp": "mg_collect_circles",
            "k_circle_labels": "mg_circle_labels", "k_window_order": "mg_roi_window_order", "k_roi": "mg_roi_segment_reduce", "k_counts_to_offsets": "mg_roi_segment_reduce"}


def agg(d, name):
    out = collections.defaultdict(float)
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != name:
            continue
        for pat, st in STAGE_OF.items():
            if pat in r["Kernel_Name"]:
                out[st] += float(r["Counter_Value"])
                break
    return out


f, w = agg(fetch_dir, "FETCH_SIZE"), agg(write_dir, "WRITE_SIZE")
res = {}
for st in sorted(set(f) | set(w)):
    fetch = f.get(st, 0.0) * 1024 * 2 / steps
    write = w.get(st, 0.0) * 1024 / steps
    res[st] = {"fetch_bytes_per_step_corrected_x2": fetch, "fetch_bytes_per_step_raw": fetch / 2,
               "write_bytes_per_step": write, "hbm_bytes_per_step": fetch + write}
print(json.dumps({"tag": tag, "source_hash": bench.source_hash(), "shape": [64, 4, 4096, 5000000], "steps_profiled": steps,
                  "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 "
                          "--no-cpu` (eager launches); FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-byte "
                          "requests at 64 bytes); KiB -> bytes; divided by the steps the process ran (warm-up, timed, the "
                          "per-stage profile pass and the result step all run the same step)",
                  "stages": res}, indent=1))
