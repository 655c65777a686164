"""Turn rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.
usage: python tools/summarize_profiles.py <tag> <kernel_stats_dir> <pmc_fetch_dir> <pmc_write_dir> <steps_in_pmc_runs>"""
import collections
import csv
import glob
import json
import shutil
import sys

tag, stats_dir, fetch_dir, write_dir, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
shutil.copy(glob.glob(f"{stats_dir}/**/*kernel_stats.csv", recursive=True)[0], f"profiles/{tag}_kernel_stats.csv")

STAGE_OF = {"k_flatfield_max": "mg_flatfield_max", "k_apply_stitch": "mg_flatfield_apply_stitch", "k_u8_blur": "mg_to_uint8_blur", "k_blur_hist": "mg_to_uint8_blur_hist",
            "k_scharr_hist": "mg_scharr_hist", "k_canny_nms": "mg_canny_nms", "k_hysteresis": "mg_canny_hysteresis",
            "k_cell_": "mg_edge_grid", "k_edge_angles": "mg_edge_angles", "k_candidates": "mg_candidate_circles",
            "k_layer_": "mg_bitmap_to_circles", "k_tile_": "mg_bitmap_to_circles", "k_score_tiles": "mg_score_circles",
            "k_prefilter": "mg_score_circles", "k_exact": "mg_score_circles", "k_nms<": "mg_nms_rounds",
            "k_collect": "mg_collect_circles", "k_clamp": "mg_collect_circles", "k_circle_labels": "mg_circle_labels",
            "k_window_order": "mg_roi_window_order", "k_roi": "mg_roi_segment_reduce"}


def agg(d, name):
    out = collections.defaultdict(float)
    for r in csv.DictReader(open(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0])):
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
    fetch = f.get(st, 0.0) * 1024 * 2 / steps  # gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
    write = w.get(st, 0.0) * 1024 / steps
    res[st] = {"fetch_bytes_per_step_corrected_x2": fetch, "fetch_bytes_per_step_raw": fetch / 2,
               "write_bytes_per_step": write, "hbm_bytes_per_step": fetch + write}
    print(f"{st:34s} fetch(x2) {fetch / 1e9:7.2f} GB  write {write / 1e9:7.2f} GB")
sys.path.insert(0, ".")
import bench  # noqa: E402  (source_hash: the file is only trusted while the sources are the ones it was measured on)

json.dump({"workload": "C4 64x4x4096x4096 u16, mode P, num_iter 5e6 (bench.py defaults)",
           "shape": [64, 4, 4096, 5000000], "source_hash": bench.source_hash(),
           "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE doubled per "
                   "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); KB -> bytes; per bench step",
           "stages": res}, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
