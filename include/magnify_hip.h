/*
 * magnify_hip.h -- C ABI of the MI355X (gfx950) marker-detection hot path.
 *
 * The reference (FordyceLab/magnify v0.12.5) is pure Python; its compiled work on
 * this path comes from OpenCV, numba-JIT helpers and NumPy.
 * There is no FFI in the reference: each entry point below replaces one of those
 * native call sites (cited as file:line relative to the reference root) and is
 * what a ctypes binding in the reference would call.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer owned by the caller (PyTorch-ROCm
 *     tensors in this repo); the library never allocates or frees device memory;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*) and is
 *     asynchronous with respect to the host;
 *   - return value: 0 = ok, MG_EINVAL = bad argument, MG_ELAUNCH = HIP launch error;
 *   - "plane" = one (channel, time) image of H x W pixels; kernels are batched over
 *     planes (gridDim.z / gridDim.y) and read per-plane counters from device memory
 *     so that no host round trip is needed between stages;
 *   - thread-safe as long as streams and buffers are distinct.
 */
#ifndef MAGNIFY_HIP_H
#define MAGNIFY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_EINVAL (-1)
#define MG_ELAUNCH (-2)
#define MG_EIO (-3) /* mg_host_read_runs only */

/* element types of image/tile buffers */
#define MG_U8 0
#define MG_U16 1
#define MG_F32 2
#define MG_F64 3

/* value stored in the angle map for "not an edge pixel" (angles are in [-pi, pi]) */
#define MG_NO_EDGE 100.0f
/* score written for circles that the exact prefilter proved to be below min_roundness */
#define MG_SCORE_SKIPPED (-2.0f)

/* Canny map values (OpenCV's): 0 weak candidate, 1 not an edge, 2 strong edge */

int mg_version(void);

/* ------------------------------------------------------------------------------------
 * Host-side tables (no GPU needed).  utils.py:433-465 circle_points,
 * utils.py:398-430 filled_circle_points, utils.py:38 cv.circle(thickness=-1).
 * ---------------------------------------------------------------------------------- */

/* Perimeter offsets (row, col) in the reference's emission order.  Returns the number
 * of points (writes at most cap of them; pass cap=0 to query). */
int mg_circle_points(int r, int four_connected, int32_t* out_rc, int cap);
/* Filled disk of filled_circle_points(r) as per-row half widths: out[dy + r] = max |dx|
 * for dy in [-r, r].  r >= 2 (undefined below that in the reference).  Returns 2r+1. */
int mg_disk_halfwidths(int r, int32_t* out);
/* Filled disk of cv.circle(..., thickness=-1): out[|dy|] = max |dx|, |dy| in [0, r]. */
int mg_cv_disk_halfwidths(int r, int32_t* out);
/* Concatenated perimeter tables for radii min_r..max_r: offsets (row, col) int32,
 * expected angle atan2(row, col) float64 (utils.py:234), starts[nr+1].  Returns the
 * total number of points, or the required capacity if cap is too small. */
int mg_perimeter_table(int min_r, int max_r, int32_t* out_rc, double* out_expected, int32_t* out_starts, int cap);

/* ------------------------------------------------------------------------------------
 * A2 flat-field (preprocess.py:83-87) and A1 stitch (stitch.py:22-39)
 * ---------------------------------------------------------------------------------- */

/* Pass 1: the two global maxima M1 = max(clip(x - dark, 0)) and M2 = max(clip(x - dark, 0) / flat)
 * (float64, NaN-propagating) of each of n_groups equal groups of tiles (n_groups = 1: the
 * reference's single-assay semantics, the maxima span the whole array; n_groups = number of
 * assays when every time slice is its own assay).  d_max2 is double[n_groups][2], pre-initialised
 * by the caller to -inf.  dark/flat: scalar when d_dark / d_flat is NULL, else a (ty, tx) image
 * of type dark_dtype / flat_dtype (MG_F32 or MG_F64) broadcast over tiles.
 * d_scratch (optional): mg_flatfield_max_scratch_floats(dtype, ty, tx) floats owned by the caller and filled by
 * mg_flatfield_bound for THIS flat image (float32, integer pixel type `dtype`; -1 / MG_EINVAL where there is no such
 * path: the tile size must be a multiple of 8 (16) pixels): a per-chunk bound an eighth (a sixteenth) of the image's
 * size.  With it the integer-pixel / float32-flat path touches the flat image itself only where a pixel can still raise
 * M2 (same result, less traffic); the bound is computed once per flat image, not once per call. */
int64_t mg_flatfield_max_scratch_floats(int dtype, int ty, int tx);
int mg_flatfield_bound(const void* d_flat, int flat_dtype, int dtype, int ty, int tx, float* d_scratch,
                       int64_t scratch_floats, void* stream);
int mg_flatfield_max(const void* d_tiles, int dtype, int64_t n_tiles, int n_groups, int ty, int tx,
                     double dark, const void* d_dark, int dark_dtype,
                     double flat, const void* d_flat, int flat_dtype,
                     double* d_max2, float* d_scratch, int64_t scratch_floats, void* stream);

/* 1 when the correction is the identity for this pixel type: integer pixels, scalar dark 0 and scalar flat 1 (the
 * defaults of the reference's flatfield_correct, preprocess.py:62).  mg_flatfield_apply_stitch then only crops and
 * copies, and pass 1 (mg_flatfield_max) is not needed. */
int mg_flatfield_is_identity(int dtype, double dark, const void* d_dark, double flat, const void* d_flat);

/* Pass 2, fused with the stitch crop/concat: out[p, R*hy, Cc*hx] in the input dtype,
 * value = trunc(((clip(x - dark, 0) / flat) * M1) / M2) with M1, M2 = d_max2[p / planes_per_group].
 * Tiles are laid out (plane, tile_row, tile_col, ty, tx); hy = ty - overlap etc.
 * If apply_flatfield == 0 this is the pure stitch copy (any dtype, d_max2 unused).
 * d_minmax (optional, double[n_planes][2] pre-initialised to {+inf, -inf}) receives the
 * per-plane min/max of the values written (feeds to_uint8, utils.py:24-26). */
int mg_flatfield_apply_stitch(const void* d_tiles, int dtype, int64_t n_planes, int n_tile_rows, int n_tile_cols,
                              int ty, int tx, int overlap, int apply_flatfield, int planes_per_group,
                              double dark, const void* d_dark, int dark_dtype,
                              double flat, const void* d_flat, int flat_dtype,
                              const double* d_max2, void* d_image, double* d_minmax, void* stream);

/* Per-plane min/max (utils.py:24-25) of strided planes.  d_minmax double[n_planes][2],
 * pre-initialised to {+inf, -inf}.  Strides are in elements. */
int mg_plane_minmax(const void* d_src, int dtype, int n_planes, int64_t plane_stride, int h, int w,
                    int64_t row_stride, double* d_minmax, void* stream);

/* ------------------------------------------------------------------------------------
 * A3-A6 edge stage of find_circles (utils.py:20-27, 115-142)
 * ---------------------------------------------------------------------------------- */

/* to_uint8 (global min/max rescale, truncating) fused with cv.GaussianBlur(5x5, sigma 0):
 * d_blur[n_planes][h][w] u8.  d_u8 (optional, same shape) receives the un-blurred uint8
 * image.  With dtype == MG_U8 and d_minmax == NULL the input is taken as already uint8. */
int mg_to_uint8_blur(const void* d_src, int dtype, int n_planes, int64_t plane_stride, int h, int w,
                     int64_t row_stride, const double* d_minmax, uint8_t* d_blur, uint8_t* d_u8, void* stream);

/* Scharr gradients of the blurred image and a histogram of the integer squared
 * magnitude m = dx^2 + dy^2 (the float32 gradient of utils.py:120 is sqrt(float(m)), a
 * monotone function of m, so np.quantile's order statistics are recovered exactly).
 * mode 0 (combined, 12288 bins per plane): bins [0, 8192) count m exactly, bin 8192 + (m >> 13)
 *   counts the rest coarsely -- one pass suffices whenever both quantile ranks fall below 8192;
 * mode 1 (window, 8192 bins per plane): bin m - d_base[plane] for m in [base, base + 8192); a plane whose base
 *   is 0xFFFFFFFF is skipped.
 * d_hist must be pre-zeroed. */
int mg_scharr_hist(const uint8_t* d_blur, int n_planes, int h, int w, int mode, const uint32_t* d_base,
                   uint32_t* d_hist, uint32_t* d_scratch, int64_t scratch_words, void* stream);
/* Words of caller-owned scratch that let mg_scharr_hist hand its per-workgroup histograms over with
 * plain stores (summed by a second small kernel) instead of one global atomicAdd per non-empty bin.
 * d_scratch == NULL selects the atomic hand-over. */
int64_t mg_scharr_hist_scratch_words(int n_planes, int h, int w, int mode);

/* mg_to_uint8_blur and mg_scharr_hist (mode 0) of the same planes as ONE call (utils.py:20-27 and 115-125 follow each
 * other in find_circles): where the input is an integer type, no un-blurred copy is asked for, w % 4 == 0 and the
 * rows are aligned, one kernel blurs a strip and histograms its Scharr magnitudes while the blurred rows are still
 * in registers; otherwise the two passes run one after the other.  Same results either way.  d_hist must be
 * pre-zeroed; d_scratch: mg_blur_hist_scratch_words words (NULL: the two passes, histogram handed over by atomics). */
int mg_to_uint8_blur_hist(const void* d_src, int dtype, int n_planes, int64_t plane_stride, int h, int w,
                          int64_t row_stride, const double* d_minmax, uint8_t* d_blur, uint8_t* d_u8, uint32_t* d_hist,
                          uint32_t* d_scratch, int64_t scratch_words, void* stream);
int64_t mg_blur_hist_scratch_words(int n_planes, int h, int w);

/* np.quantile(grad, q) for the two Canny quantiles and cv::Canny's threshold preparation (utils.py:126-134),
 * from the combined histogram of mg_scharr_hist (mode 0), on the device: ranks4 (HOST array) = the prev / next
 * order-statistic indices of numpy's linear interpolation for the low and the high quantile, gamma_* its
 * float32 weights (magnify_amd.hotpath.quantile_indexes).  Outputs d_thresh[n_planes][2] (the integer
 * thresholds mg_canny_nms takes), d_quantiles[n_planes][2] (float32, what np.quantile returns) and
 * d_unresolved[n_planes]: low byte = the number of WINDOW passes the plane still needs (0: thresholds valid) -- one
 * per distinct coarse bin that one of its ranks fell into --, next byte = the number it needs in all.  With d_state (int32 [n_planes][16]) and d_win_base
 * (uint32 [n_planes]) the passes stay on the device: d_win_base receives the base of pass 0 (0xFFFFFFFF: the plane
 * needs none and mg_scharr_hist mode 1 skips it); then, for pass = 0, 1, ...: clear d_hist_win
 * [n_planes][8192], mg_scharr_hist(mode 1, d_base = d_win_base, d_hist = d_hist_win),
 * mg_edge_thresholds_window(pass) -- which resolves the ranks of that window, writes the next base and, after a
 * plane's last window, its thresholds and d_unresolved = 0.  At most 4 passes.  Both NULL: detection only. */
int mg_edge_thresholds(const uint32_t* d_hist, int n_planes, const int64_t* ranks4, float gamma_low, float gamma_high,
                       int32_t* d_thresh, float* d_quantiles, int32_t* d_unresolved, int32_t* d_state,
                       uint32_t* d_win_base, void* stream);
int mg_edge_thresholds_window(const uint32_t* d_hist_win, int n_planes, int pass, float gamma_low, float gamma_high,
                              int32_t* d_state, uint32_t* d_win_base, int32_t* d_thresh, float* d_quantiles,
                              int32_t* d_unresolved, void* stream);

/* Bitmaps over pixels use the linear layout bit i of word k <-> pixel 32 k + i (i = y * w + x);
 * words_per_plane >= ceil(h w / 32) + 1. */

/* cv.Canny(dx, dy, L2gradient=True) non-maximum suppression + double threshold with the
 * already prepared integer thresholds d_thresh[n_planes][2] = {low, high}.  Output: two bitmaps,
 * d_weak (local maxima with m > low: OpenCV map values 0 and 2) and d_strong (m > high: value 2).
 * d_class (optional, [n_planes][3][words_per_plane]): bit planes c0, c1, c2 of every pixel's gradient
 * orientation bin floor((atan2(dy, dx) mod pi) / (pi / 8)) = 4 c1 + 2 c0 + c2, decided exactly on the integer
 * Scharr gradient: the quarter 2 c1 + c0 (same sign: |dy| < |dx| -> 0 else 1; opposite sign: |dy| > |dx| -> 2
 * else 3) and its upper half c2 (tan(pi/8) = sqrt(2) - 1 and tan(3 pi/8) = sqrt(2) + 1 as integer
 * inequalities (|dx| + |dy|)^2 > 2 dx^2, (|dy| - |dx|)^2 > 2 dx^2); consumed by the scoring prefilters
 * (mg_score_circles: quarters; mg_score_circles_keyed: eighths).
 * The words behind the last pixel of a plane are never written: the caller zeroes the bitmaps once, when it
 * allocates them (the words inside the image are overwritten by every call). */
int mg_canny_nms(const uint8_t* d_blur, int n_planes, int h, int w, const int32_t* d_thresh, uint32_t* d_weak,
                 uint32_t* d_strong, uint32_t* d_class, int64_t words_per_plane, void* stream);

/* One sweep of 8-connected hysteresis, bit-parallel on the bitmaps: every 256 x 256 tile grows its strong set into
 * its weak set to a fixed point in LDS (halo from global memory) and ORs the new bits into d_strong.  Where that
 * growth reaches a weak, not yet strong pixel of a NEIGHBOURING tile, the neighbour's flag in d_flags_out
 * (uint8[n_planes][tiles_y][tiles_x], mg_hysteresis_tiles; pre-zeroed) is set and d_changed[n_planes] becomes
 * non-zero for the plane: another sweep is needed, in which only the flagged tiles (d_flags_in = that d_flags_out)
 * are worked on.  Call until a sweep leaves d_changed at zero: d_strong is then the edge map of utils.py:142.
 * d_flags_in = NULL (first sweep, or a caller without flags): every tile is worked on. */
int mg_canny_hysteresis(const uint32_t* d_weak, uint32_t* d_strong, int64_t words_per_plane, int n_planes, int h,
                        int w, uint32_t* d_changed, const uint8_t* d_flags_in, uint8_t* d_flags_out, void* stream);
int mg_hysteresis_tiles(int h, int w, int* tiles_x, int* tiles_y);

/* Inspection helper: bitmap -> {0,1} bytes, d_out[n_planes][n_bits]. */
int mg_unpack_bits(const uint32_t* d_bits, int64_t words_per_plane, int n_planes, int64_t n_bits, uint8_t* d_out,
                   void* stream);

/* grid_array (utils.py:347-377) from the bitmap.  phases & 1: d_cell_counts / d_cell_starts [n_planes][gr*gc]
 * (gr = ceil(h / grid), gc = ceil(w / grid)) and d_num_edges[n_planes]; phases & 2: the cell-major /
 * row-major-inside-cell list d_coords[n_planes][coord_cap][2] (row, col) is filled.  A caller without an estimate of
 * the edge count runs phase 1, reads d_num_edges and sizes the list; one with a capacity from an earlier call runs
 * both at once (phases = 3): a plane with more edges than coord_cap then gets d_num_edges = 0 -- nothing downstream
 * indexes beyond the list -- and its true count in d_edge_totals[n_planes] (optional), for the caller to check when it
 * next synchronises.  d_scan_state (optional): mg_edge_grid_scan_words(...) 64-bit words owned by the caller
 * (cleared by the call itself): the prefix sum over the cells then runs on many workgroups per plane, chunk totals
 * handed on through these words, chunks taken by ticket (no assumption about the order workgroups are dispatched in). */
int64_t mg_edge_grid_scan_words(int n_planes, int h, int w, int grid);
int mg_edge_grid(const uint32_t* d_edge_bits, int64_t words_per_plane, int n_planes, int h, int w, int grid,
                 int32_t* d_cell_counts, int32_t* d_cell_starts, int32_t* d_num_edges, int32_t* d_coords,
                 int64_t coord_cap, uint64_t* d_scan_state, int32_t* d_edge_totals, int phases, void* stream);

/* float32 gradient angle arctan2(dy, dx) (utils.py:118-119, 170) at every edge pixel of the
 * compact list, evaluated in float64 and rounded once: d_angle[n_planes][h][w] is written at
 * edge pixels only. */
int mg_edge_angles(const uint8_t* d_blur, int n_planes, int h, int w, const int32_t* d_coords, int64_t coord_cap,
                   const int32_t* d_num_edges, float* d_angle, void* stream);

/* ------------------------------------------------------------------------------------
 * A8-A11 RANSAC circle candidates, scoring, greedy NMS (utils.py:145-199, 225-344)
 * ---------------------------------------------------------------------------------- */

/* Layout of the circle de-duplication bitmap: centres are binned into 64 x 64 tiles of the padded
 * centre grid (row + max_r, col + max_r); every (tile, radius) pair owns one 4096-bit "layer"
 * (bit = (row_in_tile << 6) | col_in_tile); layers are ordered (tile_row, tile_col, r).
 * n_layers = tile_rows * tile_cols * (max_r - min_r + 1), bitmap_words = 128 * n_layers. */
int mg_dedup_layout(int h, int w, int min_r, int max_r, int* n_tile_rows, int* n_tile_cols, int64_t* n_layers,
                    int64_t* bitmap_words);

/* candidate_circles (utils.py:295-344) with the build's counter-based RNG (the reference
 * is unseeded): iteration i of plane p draws three 32-bit uniforms from
 * splitmix64(seed[p] + (3 i + k + 1) * golden) >> 32.  p0 is a jittered stratified draw: iteration i
 * owns the slice [a, b) = [floor(i E / K), floor((i + 1) E / K)) of the cell-major edge list and
 * takes p0 = coords[a + (u0 * max(b - a, 1) >> 32)] (every edge equally likely, consecutive
 * iterations spatially coherent); p1, p2 = p0's cell list[u * count >> 32].  Then steps 4 of filter_circles
 * (utils.py:157-166): radius window, round-half-even, off-image rejection.  Survivors set
 * their bit in d_bitmap[n_planes][bitmap_words] (mg_dedup_layout), which de-duplicates them:
 * a circle's score depends only on (row, col, r).
 * d_raw (optional, float32 [n_planes][num_iter][3]) receives the unfiltered circles. */
int mg_candidate_circles(const int32_t* d_coords, int64_t coord_cap, const int32_t* d_cell_starts,
                         const int32_t* d_cell_counts, const int32_t* d_num_edges, int n_planes, int h, int w,
                         int grid, const uint64_t* d_seeds, int64_t num_iter, int min_r, int max_r,
                         uint32_t* d_bitmap, int64_t bitmap_words, float* d_raw, void* stream);

/* The same candidates WITHOUT global atomics: every iteration stores a 32-bit key into
 * d_keys[n_planes][num_iter] -- (tile << 17) | ((r - min_r) << 12) | ((pr & 63) << 6) | (pc & 63) with
 * (pr, pc) = (row + max_r, col + max_r), tile = (pr >> 6) * tile_cols + (pc >> 6) -- or 0xFFFFFFFF when
 * step 4 of filter_circles rejects it.  Needs tile_rows * tile_cols < 32768 and max_r - min_r < 32
 * (MG_EINVAL otherwise: use mg_candidate_circles). */
int mg_candidate_keys(const int32_t* d_coords, int64_t coord_cap, const int32_t* d_cell_starts,
                      const int32_t* d_cell_counts, const int32_t* d_num_edges, int n_planes, int h, int w, int grid,
                      const uint64_t* d_seeds, int64_t num_iter, int min_r, int max_r, uint32_t* d_keys, float* d_raw,
                      void* stream);

/* Keys -> unique keys, grouped by tile.  Because p0 is a stratified draw over the cell-major edge
 * list, the iterations that can place a centre in a given 64 x 64 tile form one contiguous key range
 * per cell row within max_r + 2 of the tile; one workgroup per pair of tiles scans them, ORs its own
 * keys into an LDS copy of the tiles' layers, counts, reserves its slice of the plane's list with one
 * atomicAdd on d_num_circles (zeroed by this call unless counters_clear, see below) and stores the unique keys of each tile in
 * (r, row, col) order:
 *   d_unique_keys[n_planes][circle_cap] uint32, d_tile_ranges[n_planes][n_tiles][2] int32 = (first, count)
 *   of every tile (n_tiles = tile rows x tile cols of mg_dedup_layout), d_num_circles[n_planes].
 * The slices of different workgroups land in arrival order: list positions are not canonical, the
 * keys are (mg_nms_rounds / mg_collect_circles take them as tie-breakers).  d_cell_* / d_num_edges /
 * grid / num_iter must be those the keys were generated with.
 * d_layer_starts (optional, [n_planes][n_tiles][nr + 1], nr = max_r - min_r + 1): list index of the first key
 * of every radius of every tile, entry nr = the tile's end (mg_score_circles_keyed deals its work by radius).
 * counters_clear (here and in mg_score_circles_keyed, mg_nms_rounds, mg_collect_circles): 0 -- the call clears the
 * per-plane counters it accumulates into (d_num_circles / d_num_surv / d_undecided / d_num_out) with a launch of its
 * own; 1 -- the caller has cleared them (a caller that keeps all its counters in one block clears it once per chain:
 * eight ~5 us launches fewer, a tenth of a single-plane call). */
int mg_keys_to_circles(const uint32_t* d_keys, int64_t num_iter, const int32_t* d_cell_starts,
                       const int32_t* d_cell_counts, const int32_t* d_num_edges, int n_planes, int h, int w, int grid,
                       int min_r, int max_r, uint32_t* d_unique_keys, int64_t circle_cap, int32_t* d_tile_ranges,
                       int32_t* d_num_circles, int32_t* d_layer_starts, int counters_clear, void* stream);

/* Ordered compaction of the bitmap into the unique circle list in the build's canonical order
 * (tile_row, tile_col, r, row, col): d_circles[n_planes][circle_cap][3] int32 (row, col, r),
 * d_num_circles[n_planes], and d_layer_offsets[n_planes][n_layers + 1] (start of every layer in
 * the list; the last entry is the total).  Clears the bitmap words it consumes (the bitmap is
 * all-zero again afterwards). */
int mg_bitmap_to_circles(uint32_t* d_bitmap, int64_t bitmap_words, int n_planes, int h, int w, int min_r,
                         int max_r, int32_t* d_layer_offsets, int32_t* d_circles, int64_t circle_cap,
                         int32_t* d_num_circles, void* stream);

/* mean_grad / len(perimeter) (utils.py:183-188, 225-251), one workgroup per centre tile with the
 * tile's window of the 1-bit edge map staged in LDS.
 * Pass A (exact prefilter): every term of the sum is <= 1, so circles with fewer than
 * min_roundness * P edge pixels on their perimeter cannot pass (they get MG_SCORE_SKIPPED when
 * write_skipped != 0, else their score is left unwritten).  With d_class_bits (optional, from
 * mg_canny_nms) an edge pixel only counts where its gradient orientation class is not the one
 * perpendicular to the perimeter point's radial direction: such a pixel is at least pi/4 away from
 * radial, its term 4 |d - pi/2| / pi - 1 is <= 0, and the bound stays exact.
 * Pass B: the reference's float64 sum, sequential in perimeter order, stored float32 and divided
 * by the perimeter length in float32; d_angle holds the gradient angle at edge pixels
 * (mg_edge_angles); per_total = number of entries of the perimeter tables.
 * Circles with score >= min_roundness (float32 compare, utils.py:191) are appended (unordered) to
 * d_alive[n_planes][circle_cap] (indices into d_circles), d_num_alive[n_planes] pre-zeroed;
 * d_max_rc[n_planes][2] (pre-set to INT32_MIN) receives max row / max col of the alive circles
 * (the claim-grid extent of utils.py:268-270).  d_num_scored (optional, [n_planes], pre-zeroed)
 * counts the circles that reached pass B.
 * Input, one of: (a) d_unique_keys + d_tile_ranges from mg_keys_to_circles (d_layer_offsets unused):
 * the circles are decoded from the keys and d_circles[n_planes][circle_cap][3] is an OUTPUT, written
 * only at the positions of the circles that pass the threshold (all that suppression and the ordered
 * output read); (b) d_unique_keys == NULL: d_circles + d_layer_offsets from mg_bitmap_to_circles.
 * dedup_centres != 0 (only when greedy suppression with min_dist > 0 follows): of the passing circles that
 * share a centre, only the first in suppression order (score desc, radius asc) is appended to d_alive --
 * the others have the same suppression ring and are rejected whatever happens to the first one
 * (utils.py:254-292), so the kept set is unchanged; best effort per tile (64 parked circles). */
int mg_score_circles(const float* d_angle, const uint32_t* d_edge_bits, const uint32_t* d_class_bits,
                     int64_t words_per_plane, int n_planes, int h, int w, int32_t* d_circles, int64_t circle_cap,
                     const int32_t* d_layer_offsets, const uint32_t* d_unique_keys, const int32_t* d_tile_ranges,
                     int min_r, int max_r, const int32_t* d_per_rc, const double* d_per_expected,
                     const int32_t* d_per_starts, int per_total, float min_roundness, int write_skipped,
                     int dedup_centres, float* d_scores, int32_t* d_alive, int32_t* d_num_alive, int32_t* d_max_rc,
                     int32_t* d_num_scored, void* stream);

/* The keyed path's scoring (same scores and same passing set as mg_score_circles input (a), which remains for
 * radii outside mg_score_keyed_supported): two kernels.
 * Prefilter: a workgroup per super-tile of 2 x 4 centre tiles (128 x 256 positions) with the edge window in LDS as one byte per
 * pixel (the gradient-orientation bin of mg_canny_nms' class planes, 0x0C = no edge); a lane per circle, all
 * circles of a wave of one radius, the perimeter walked as straight-line code per radius.  For every pair of
 * opposite perimeter points the two window bytes select, in one byte permute, UPPER BOUNDS of the two pixels'
 * terms 4 |d - pi/2| / pi - 1 (utils.py:244-249) from the pair's table (mg_score_pair_table: the distance
 * between the points' radial direction and the pixel's orientation bin bounds the term); the bounds are summed
 * in 1/64 (rounded up) and a circle whose bound is below min_roundness * P - 1e-3 is dropped -- exact: every
 * term <= its bound.  Survivors are appended to d_surv_list[n_planes][surv_cap][2] (scratch: list index and key;
 * surv_cap >= circle_cap is required: it can then never overflow), d_num_surv[n_planes] (scratch, zeroed here).
 * Exact pass: the reference's float64 sum in perimeter order, 4 (16 at small batches) lanes per survivor (the edge
 * pixels on the perimeter and their angles are found in parallel, the terms added in order); the gradient angle of a
 * hit is d_angle's entry or, with d_angle == NULL, computed on demand from d_blur exactly as mg_edge_angles
 * does (so neither the angle map nor its pass is needed).  Outputs as mg_score_circles (no same-centre
 * reduction: the suppression treats same-centre circles correctly by itself); d_num_scored = survivors.
 * d_class_bits is required. */
int mg_score_keyed_supported(int min_r, int max_r); /* 1: 2 <= min_r, max_r <= 26 */
/* out_entries[27][80]: for radius r and pair k of opposite perimeter points, 8 signed bytes = bound (1/64) per
 * orientation bin.  Returns the number of entries (2160); fills them when cap >= that. */
int mg_score_pair_table(uint64_t* out_entries, int cap);
/* first point (dr, dc) of every pair of radius r, in table order; returns the number of pairs */
int mg_score_pairs(int r, int32_t* out_rc, int cap);
int mg_score_circles_keyed(const uint8_t* d_blur, const float* d_angle, const uint32_t* d_edge_bits,
                           const uint32_t* d_class_bits, int64_t words_per_plane, int n_planes, int h, int w,
                           int32_t* d_circles, int64_t circle_cap, const uint32_t* d_unique_keys,
                           const int32_t* d_layer_starts, int min_r, int max_r, const int32_t* d_per_rc,
                           const double* d_per_expected, const int32_t* d_per_starts, int per_total,
                           const uint64_t* d_pair_table, float min_roundness, int write_skipped, float* d_scores,
                           int32_t* d_alive, int32_t* d_num_alive, int32_t* d_max_rc, int32_t* d_num_scored,
                           int32_t* d_surv_list, int64_t surv_cap, int32_t* d_num_surv, int counters_clear,
                           void* stream);

/* n_rounds rounds of the parallel-but-equivalent greedy suppression of filter_neighbors
 * (utils.py:254-292).  Priority = (score desc, tie key asc) -- the build's canonical tie order
 * (tile_row, tile_col, r, row, col): d_tie_keys[n_planes][circle_cap] (the unique keys of
 * mg_keys_to_circles) or, when NULL, the index in d_circles (canonical after mg_bitmap_to_circles).
 * d_grid[n_planes][grid_cap] uint64 claim grid pre-set to all-ones;
 * d_state[n_planes][circle_cap] uint8 (0 undecided, 1 kept, 2 dropped) pre-zeroed for
 * alive circles; round k sets d_undecided[k * undecided_stride + plane] to 0 (counters_clear: the caller has), or to
 * 1 when a circle of the plane is still undecided after it: the rounds have converged when a round leaves none.
 * Ring = 4-connected perimeter of radius min_dist (d_ring_rc, ring_len); indices wrap
 * modulo the claim grid extent like negative numba indices do. */
/* max_alive (all suppression calls): an upper bound of d_num_alive known to the caller, used only to size the
 * launch grid (0 = unknown: a grid for circle_cap).  d_skip (all three; may be NULL): int32[n_planes], planes with a
 * non-zero entry are left alone -- they were decided by mg_nms_sparse. */
int mg_nms_rounds(const int32_t* d_circles, int64_t circle_cap, const float* d_scores, const int32_t* d_alive,
                  const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes, int min_dist,
                  const int32_t* d_ring_rc, int ring_len, uint64_t* d_grid, int64_t grid_cap, uint8_t* d_state,
                  int32_t* d_undecided, int64_t undecided_stride, int n_rounds, int counters_clear,
                  const uint32_t* d_tie_keys, int64_t max_alive, const int32_t* d_skip, void* stream);

/* Before the rounds (optional, exact): of the alive circles that share a centre only the first in suppression
 * order (score desc, tie key asc) stays undecided, the others are marked rejected -- they have the same ring and
 * are rejected whatever happens to the first (utils.py:254-292).  One bid on the centre's own claim-grid cell,
 * which is restored before returning; d_state as for mg_nms_rounds. */
int mg_nms_same_centre(const int32_t* d_circles, int64_t circle_cap, const float* d_scores, const int32_t* d_alive,
                       const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes, int min_dist, uint64_t* d_grid,
                       int64_t grid_cap, uint8_t* d_state, const uint32_t* d_tie_keys, int64_t max_alive,
                       const int32_t* d_skip, void* stream);

/* After the rounds have converged: restore the all-ones claim grid under the rings of all alive
 * circles, so that the grid needs its full initialisation only once. */
int mg_nms_cleanup(const int32_t* d_circles, int64_t circle_cap, const float* d_scores, const int32_t* d_alive,
                   const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes, int min_dist,
                   const int32_t* d_ring_rc, int ring_len, uint64_t* d_grid, int64_t grid_cap, uint8_t* d_state,
                   int64_t max_alive, const int32_t* d_skip, void* stream);

/* The same greedy suppression (utils.py:254-292) decided from the circles alone, one workgroup per plane: circle i is
 * kept iff no circle j kept before it has a ring cell in common with it, i.e. iff c_i - c_j is in D = ring (-) ring for
 * no such j.  d_dbits: the bitmap of D, bit (dr + 2 d)(4 d + 1) + dc + 2 d set iff two rings of radius d = min_dist
 * whose centres differ by (dr, dc) share a cell (the caller builds it from the ring of mg_circle_points).  A plane's
 * alive circles are bucketed in LDS, a circle looks at the buckets around it; d_state gets 1 (kept) / 2 (dropped) for
 * every alive circle of a plane it decides and d_done[plane] = 1.  d_done[plane] = 0 -- the plane is left to
 * mg_nms_same_centre / mg_nms_rounds, untouched -- when it holds more than 12 288 alive circles, spans more than 9 216
 * buckets of 32 x 64, or a centre lies at row or col < -(min_dist + 1) (the reference's claim index would be negative
 * and wrap); every plane when min_dist > mg_nms_sparse_max_dist().  d_skip of the three calls above = this d_done:
 * they leave the planes alone that are decided (the cleanup only zeroes their circles' state bytes).  Needs no claim
 * grid and does not look at d_state before writing it. */
int mg_nms_sparse(const int32_t* d_circles, int64_t circle_cap, const float* d_scores, const int32_t* d_alive,
                  const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes, int min_dist, const uint32_t* d_dbits,
                  uint8_t* d_state, const uint32_t* d_tie_keys, int32_t* d_done, void* stream);
int mg_nms_sparse_max_dist(void);

/* Gather the kept circles in priority order (utils.py:195-199 output order):
 * d_out[n_planes][out_cap][3] int32 (row, col, r), d_out_scores, d_num_out[n_planes].
 * keep_all != 0 skips the state test (min_dist == 0: no suppression, utils.py:197).
 * d_scratch: int32[n_planes][3 * out_cap] work space (the kept circles' indices and 64-bit priority keys). */
int mg_collect_circles(const int32_t* d_circles, int64_t circle_cap, const float* d_scores, const int32_t* d_alive,
                       const int32_t* d_num_alive, const uint8_t* d_state, int keep_all, int n_planes,
                       int32_t* d_out, float* d_out_scores, int64_t out_cap, int32_t* d_num_out,
                       int32_t* d_scratch, const uint32_t* d_tie_keys, int counters_clear, void* stream);

/* ------------------------------------------------------------------------------------
 * A12-A15, A18 labels, ROI gather, fg/bg masks, masked reductions
 * (utils.py:380-395, 60-80; find.py:561-602; README.md:21-22, identify.py:76-80)
 * ---------------------------------------------------------------------------------- */

/* circle_labels as a coverage count: d_labels[n_planes][h][w] int32 pre-set to -1;
 * beads d_beads[n_planes][bead_cap][3] (row, col, r), d_num_beads[n_planes];
 * d_halfwidths[(max_r+1)][2*max_r+1] from mg_disk_halfwidths (row r of the table).
 * reset != 0 writes -1 back under the same disks instead (the map is clean again for reuse). */
int mg_circle_labels(const int32_t* d_beads, int64_t bead_cap, const int32_t* d_num_beads, int n_planes, int h,
                     int w, const int32_t* d_halfwidths, int max_r, int32_t* d_labels, int reset, void* stream);

/* ROI gather + masks + reductions for one assay.  image (C, T, h, w) of dtype (u8/u16/f32);
 * beads (m, 3) with labels from time 0; window = bounding_box(col, row, L, w, h).
 * Outputs: roi (m, C, T, L, L) same dtype; fg, bg (m, L, L) uint8 {0,1};
 * sums double[m][C][T][2] = {sum over fg, sum over bg} (exact for integer dtypes below 2^53),
 * counts int32[m][2] = {|fg|, |bg|}.  Any output pointer may be NULL. */
int mg_roi_gather_reduce(const void* d_image, int dtype, int n_c, int n_t, int h, int w, const int32_t* d_beads,
                         int m, int roi_len, const int32_t* d_labels, void* d_roi, uint8_t* d_fg, uint8_t* d_bg,
                         double* d_sums, int32_t* d_counts, void* stream);

/* Batched form: marker g belongs to assay d_marker_assay[g] (image base + assay * assay_stride
 * elements, labels base + assay * h * w) and owns label value d_marker_local[g].  Both index
 * arrays may be NULL (single assay, local index = g).  d_beads is (m, 3) for all markers. */
int mg_roi_gather_reduce_batched(const void* d_image, int dtype, int64_t assay_stride, int n_c, int n_t, int h, int w,
                                 const int32_t* d_beads, const int32_t* d_marker_assay,
                                 const int32_t* d_marker_local, int m, int roi_len, const int32_t* d_labels,
                                 void* d_roi, uint8_t* d_fg, uint8_t* d_bg, double* d_sums, int32_t* d_counts,
                                 void* stream);

/* d_offsets[0 .. n] = exclusive prefix sums of min(d_counts[i], cap) (int32, on the device): the d_assay_offsets of
 * mg_roi_segment_reduce from the per-assay bead counts mg_collect_circles left in device memory -- the ROI pass can
 * then be queued (with an upper bound for m) before the host has fetched the counts. */
int mg_counts_to_offsets(const int32_t* d_counts, int n, int cap, int32_t* d_offsets, void* stream);

/* The marker table a rank contributes to the final all-gather: row g of d_table[m][6 + 2 n_c] (float64)
 * = [assay_offset + assay, row, col, r, fg_count, bg_count, fg_sum[n_c], bg_sum[n_c]] of marker g, from the bead
 * tables (d_beads / bead_stride / d_assay_offsets as for mg_roi_segment_reduce: compact, or one padded row per assay)
 * and the counts (m, 2) / sums (m, n_c, n_t, 2) of the ROI pass at time index t_index.  m: the rows d_table holds; rows
 * beyond d_assay_offsets[n_assays] are left alone. */
int mg_marker_table(const int32_t* d_beads, int64_t bead_stride, const int32_t* d_assay_offsets, int n_assays, int m,
                    int assay_offset, const int32_t* d_counts, const double* d_sums, int n_c, int n_t, int t_index,
                    double* d_table, void* stream);

/* fg/bg segmentation + ROI gather + reductions WITHOUT a label map (find.py:561-602 with
 * utils.py:380-395 folded in): the masks come straight from the bead table.  A window pixel is
 * foreground iff it lies in the marker's own disk and in no other disk of its assay (labels == i),
 * background iff it lies in no disk (labels == -1).  d_beads (m, 3) = [row, col, r] of all markers,
 * assay-major; d_assay_offsets int32[n_assays + 1] delimits every assay's markers (marker g of assay
 * a has label value g - d_assay_offsets[a]).  One workgroup per marker, launched for m of them: m may be an upper
 * bound of d_assay_offsets[n_assays] (workgroups beyond it leave at once; markers beyond m are not worked on, the
 * outputs must hold m).  d_halfwidths / max_r as for mg_circle_labels (disks with
 * r < 2 or r > max_r cover nothing, as there).  bead_stride = 0: d_beads is that compact list;
 * bead_stride > 0: d_beads holds one padded row of bead_stride triples per assay (the layout
 * mg_collect_circles writes: the ROI pass can start from the device-resident tables while the host
 * is still fetching them); the outputs are compact either way.  time_major != 0: every assay's image
 * block is stored (n_t, n_c, h, w) instead of (n_c, n_t, h, w) -- a time-sharded single assay is gathered
 * where the flat-field pass left it, without a transposing copy; the outputs stay (channel, time)-ordered.
 * Outputs as mg_roi_gather_reduce. */
int mg_roi_segment_reduce(const void* d_image, int dtype, int64_t assay_stride, int n_c, int n_t, int h, int w,
                          int time_major, const int32_t* d_beads, int64_t bead_stride, const int32_t* d_assay_offsets,
                          int n_assays, int m, const int32_t* d_order, int roi_len, const int32_t* d_halfwidths, int max_r,
                          void* d_roi, uint8_t* d_fg, uint8_t* d_bg, double* d_sums, int32_t* d_counts, void* stream);

/* d_order[0 .. m) for mg_roi_segment_reduce (NULL there: markers are visited as listed): the markers of every assay
 * band by band (64 rows) and left to right inside a band, so that windows which share image lines are gathered at
 * about the same time and meet in the L2s (the reference walks its beads in score order, find.py:571-602; the order
 * of the work is free, every marker's outputs stay at the marker's index).  Tables as for mg_roi_segment_reduce. */
int mg_roi_window_order(const int32_t* d_beads, int64_t bead_stride, const int32_t* d_assay_offsets, int n_assays, int m,
                        int32_t* d_order, void* stream);

/* Masked median of an already gathered roi (m, C, T, L, L) of element type `dtype` (MG_U8 / U16 / F32 / F64):
 * roi.where(mask).median(dim=[roi_x, roi_y]) of identify.py:76-80 and filter.py:20-22, 74, 82 with numpy's nanmedian
 * semantics -- the mean of the two middle values (in float64), pixels outside the mask and NaN pixels ignored, NaN when
 * nothing is left.  The mask of marker g at timepoint t is the L x L bytes at d_mask + g * mask_stride_m +
 * t * mask_stride_t (elements; mask_stride_t = 0: one mask for all timepoints, as find_beads replicates its geometry,
 * find.py:585-586; chips searched at several timesteps have a mask per timepoint, find.py:119-140).
 * d_median double[m][C][T].  Exact: radix select on the order-preserving bit pattern of the values. */
int mg_roi_masked_median(const void* d_roi, int dtype, const uint8_t* d_mask, int64_t mask_stride_m,
                         int64_t mask_stride_t, int m, int n_c, int n_t, int roi_len, double* d_median, void* stream);
/* The same for uint16 rois under one mask (m, L, L) for all timepoints. */
int mg_roi_masked_median_u16(const uint16_t* d_roi, const uint8_t* d_mask, int m, int n_c, int n_t, int roi_len,
                             double* d_median, void* stream);

/* ------------------------------------------------------------------------------------
 * A16 / A17 ButtonFinder helpers (find.py:205-402, 632-677)
 * ---------------------------------------------------------------------------------- */

/* cluster_1d (find.py:632-677): the cost of every integer offset in [0, n_offsets) for
 * n_clusters equal-width clusters over the ascending d_sorted_points (float64):
 * sum_k var_k * sqrt(ideal_k) + penalty * (ideal_k - n_k)^2, empty clusters take the largest
 * var.  The caller takes the first minimum (strict <, find.py:666).  d_costs double[n_offsets]. */
int mg_cluster1d_costs(const double* d_sorted_points, int n_points, int n_offsets, int n_clusters,
                       double cluster_length, const double* d_ideal, double penalty, double* d_costs, void* stream);

/* fg / bg masks of find_rois (find.py:383-400): fg[g] = cv.circle(filled, radius d_radii[g]),
 * bg[g] = annulus(outer_r, inner_r), centred at d_centers[g] = (row, col) inside the
 * roi_len x roi_len window.  d_cv_halfwidths[(max_table_r + 1)][hw_stride]: row r holds
 * mg_cv_disk_halfwidths(r).  Outputs uint8 [m][roi_len][roi_len]. */
int mg_button_masks(const int32_t* d_centers, const int32_t* d_radii, int m, int roi_len, int outer_r, int inner_r,
                    const int32_t* d_cv_halfwidths, int hw_stride, int max_table_r, uint8_t* d_fg, uint8_t* d_bg,
                    void* stream);

/* Masked sums with explicit masks: roi (m, n_ct, L, L), fg / bg (m, L, L) uint8 ->
 * d_sums double[m][n_ct][2] = {fg sum, bg sum}, d_counts int32[m][2] (optional). */
int mg_masked_sums(const void* d_roi, int dtype, const uint8_t* d_fg, const uint8_t* d_bg, int m, int n_ct,
                   int roi_len, double* d_sums, int32_t* d_counts, void* stream);

/* ------------------------------------------------------------------------------------
 * Measurement aid ("confirm with a measured streaming-copy ceiling on the box")
 * ---------------------------------------------------------------------------------- */

/* One streaming pass over n_bytes (a multiple of 16, 16-byte aligned buffers) by this library's own 16-byte-per-lane
 * grid-stride kernel on `blocks` workgroups of 256 (1 .. 65535; 0: one 16-byte access per lane, n_bytes / 4096
 * workgroups): mode 0 copies d_src to d_dst, mode 1 only reads d_src (d_sink: one word, written only if a lane's
 * words XOR to one particular value -- it keeps the loads alive), mode 2 only writes d_dst.  The caller times it
 * (HIP events). */
int mg_stream_probe(const void* d_src, void* d_dst, int64_t n_bytes, int mode, uint32_t* d_sink, int blocks, void* stream);

/* ------------------------------------------------------------------------------------
 * Host side of the streamed ingest (reader.py:265-292: the reference maps every TIFF
 * page to its own dask block and tifffile reads it when the block is computed)
 * ---------------------------------------------------------------------------------- */

/* Positional reads of n byte runs -- run i = nbytes[i] bytes at offsets[i] of the open file fds[i], into dsts[i]
 * (HOST pointers; page-locked blocks in the product, so the upload that follows is asynchronous) -- by n_threads
 * threads (1 .. 64) that take runs by ticket.  No GPU work, no stream; the call returns when every run is in place.
 * Returns MG_OK; MG_EINVAL for a bad argument; MG_EIO when a read failed or met the end of its file: failed[0] = the
 * run's index, failed[1] = errno (0: end of file).  `failed` may be NULL. */
int mg_host_read_runs(const int32_t* fds, const int64_t* offsets, const int64_t* nbytes, void* const* dsts, int n,
                      int n_threads, int64_t* failed);
/* The mirror image for the results of a streamed run (accessor.py:18-35: the reference spills every assay's roi / fg / bg
 * to a zarr store; here mg.save's NetCDF writer puts a variable's bytes -- already big-endian, swapped on the device --
 * where the header says): run i = nbytes[i] bytes from srcs[i] (HOST pointers) to offsets[i] of the open file fds[i].
 * Same threads, return values and `failed` as mg_host_read_runs. */
int mg_host_write_runs(const int32_t* fds, const int64_t* offsets, const int64_t* nbytes, void* const* srcs, int n,
                       int n_threads, int64_t* failed);

#ifdef __cplusplus
}
#endif
#endif /* MAGNIFY_HIP_H */
