"""Whole-stack driver of the hot path: flat-field + stitch -> bead detection -> fg/bg
segmentation -> ROI reduction for a (time x channel x H x W) stack resident in HBM.

Detection modes:
  * mode "P" (headline): every time slice is an independent assay, which is what the reference
    does for a list input (pipeline.py:18-24): per-slice flat-field maxima, per-slice detection.
  * mode "R": the reference's single-assay semantics (find.py:477, 543-550): global maxima,
    detection on time 0 only, geometry replicated over time.

The time axis is the unit of sharding across GPUs (``magnify_amd.distributed``).
"""
from __future__ import annotations


import math
import os

import numpy as np
import torch

from . import hotpath as hp


class StackProcessor:
    """Reusable workspaces for stacks of a fixed shape (T, C, H, W)."""

    def __init__(self, n_t, n_c, h, w, dtype=torch.uint16, min_bead_diameter=10, max_bead_diameter=50,
                 low_edge_quantile=0.1, high_edge_quantile=0.9, num_iter=5_000_000, min_roundness=0.3,
                 roi_length=None, search_channels=(0,), mode="P", plane_batch=None, device="cuda", n_streams=1,
                 sub_batches=None, tile_grid=None, overlap=0):
        """``h, w``: the size of the (stitched) image.  ``tile_grid=(rows, cols, tile_y, tile_x)`` + ``overlap``:
        the input stack is (T, C, rows, cols, tile_y, tile_x) and the flat-field pass also crops and joins the
        tiles (stitch.py:22-39); ``h, w`` must then be the stitched size (``stitched_shape``)."""
        hp.require_gpu()
        if min_bead_diameter > max_bead_diameter:
            raise ValueError("min_bead_diameter must be <= max_bead_diameter.")
        self.T, self.C, self.h, self.w = n_t, n_c, h, w
        self.overlap = int(overlap)
        self.tile_grid = tuple(int(v) for v in tile_grid) if tile_grid is not None else (1, 1, h, w)
        if tile_grid is None and overlap:
            raise ValueError("overlap needs tile_grid")
        if stitched_shape(*self.tile_grid, self.overlap) != (h, w):
            raise ValueError(f"tile_grid {self.tile_grid} with overlap {overlap} stitches to "
                             f"{stitched_shape(*self.tile_grid, self.overlap)}, not to {(h, w)}")
        self.min_r = math.floor(min_bead_diameter / 2)  # find.py:461-467
        self.max_r = math.ceil(max_bead_diameter / 2)
        self.L = roi_length if roi_length is not None else 2 * max_bead_diameter
        self.low_q, self.high_q = low_edge_quantile, high_edge_quantile
        self.num_iter, self.min_roundness = num_iter, min_roundness
        self.search_channels = list(search_channels)
        self.mode = mode
        self.dev = torch.device(device)
        self.n_assays = n_t if mode == "P" else 1
        self.n_streams = n_streams if (self.n_assays >= 2 * n_streams and not plane_batch) else 1
        self.step_stats = []  # per sub-batch (device counters, host counters) of the last call
        self._roi_bound = None  # markers the ROI pass is launched for before the counts are known (None: capacity)
        self.pool_tag = ""      # which set of pooled output buffers the ROI pass writes (process_stream alternates two)
        self.placement = None   # what the placement trial of the first call measured (None: no trial)
        self._trial_blocks = None
        self._placed = False
        self.stage = None  # device staging buffer of the host-ingest path
        if self.n_streams > 1:
            # The stack is cut into contiguous sub-batches of assays; HIP stream / host thread k works
            # through sub-batches k, k + n_streams, ...: flat-field (mode P: every assay has its own
            # maxima, so sub-batches are independent) then detection.  One sub-batch's bandwidth-bound
            # flat-field and its latency-bound tails (hysteresis sweeps, suppression rounds, host
            # convergence checks) overlap the other streams' kernels.
            divisors = [d for d in range(1, self.n_assays + 1) if self.n_assays % d == 0]
            self.n_streams = max(d for d in divisors if d <= self.n_streams)
            n_sub = max([d for d in divisors if self.n_streams <= d <= (sub_batches or self.n_streams)] or [self.n_streams])
            per = self.n_assays // n_sub
            self.ranges = [(i, i + per) for i in range(0, self.n_assays, per)]
            self.streams = [torch.cuda.Stream(device=device) for _ in range(self.n_streams)]
            self.finders = [hp.CircleFinder(per, h, w, self.min_r, self.max_r, num_iter, device=device)
                            for _ in range(self.n_streams)]
            for f in self.finders:  # several host threads launching side by side: no stream captures among them
                f._graphs = None
            self.finder = self.finders[0]
            self.batch = per
        else:
            self.batch = min(self.n_assays, plane_batch or self.n_assays)
            self.finder = hp.CircleFinder(self.batch, h, w, self.min_r, self.max_r, num_iter, device=device)
        self.image = torch.empty((n_t, n_c, h, w), dtype=dtype, device=self.dev)
        self.minmax = torch.empty((n_t, n_c, 2), dtype=torch.float64, device=self.dev)

    def flatfield(self, stack: torch.Tensor, flatfield=1.0, darkfield=0.0, max2=None):
        """stack (T, C, H, W) -- or (T, C, rows, cols, tile_y, tile_x) with a tile grid -- -> self.image
        (T, C, H, W), self.minmax (T, C, 2)."""
        T, C = self.T, self.C
        tiles = stack.view(T * C, 1, *self.tile_grid)
        # mode P: every time slice is its own assay -> its own pair of maxima (n_groups = T);
        # mode R: single assay, the maxima span the whole stack (preprocess.py:84,86)
        hp.flatfield_stitch(tiles, self.overlap, flatfield, darkfield, out=self.image, minmax_out=self.minmax, max2=max2,
                            n_groups=T if self.mode == "P" else 1)
        return self.image

    def detect(self, seed=0):
        """Bead tables per assay: list of (M_a, 3) int32 [row, col, r] (find.py:475-501)."""
        T, h, w = self.T, self.h, self.w
        assays = list(range(self.n_assays))
        beads = [np.empty((0, 3), dtype=np.int32) for _ in assays]
        if self.n_streams > 1:
            return self._detect_streams(seed, beads)
        for k, ch in enumerate(self.search_channels):
            done = 0
            while done < len(assays):
                # contiguous window of `batch` assays (the last window overlaps the previous one
                # instead of being ragged; already finished assays are skipped on output)
                lo = min(done, len(assays) - self.batch)
                ids = assays[lo : lo + self.batch]
                planes = self.image[lo : lo + self.batch, ch]  # strided view, no copy
                mm = self.minmax[lo : lo + self.batch, ch].contiguous()
                seeds = [(seed + 1000003 * a + 7919 * k) & 0xFFFFFFFFFFFFFFFF for a in ids]
                res, _ = self.finder.find(planes, mm, self.low_q, self.high_q, self.min_roundness, self.min_r, seeds,
                                          stable_input=True)  # (a view of this processor's own image block)
                for j, a in enumerate(ids):
                    if a >= done:
                        beads[a] = np.concatenate([beads[a], dedup_against(beads[a], res[j][0], 2 * self.min_r)])
                done = lo + self.batch
        return beads

    def _detect_streams(self, seed, beads, stack=None, flatfield=1.0, darkfield=0.0):
        """Detection (and, when ``stack`` is given, the flat-field pass) of every sub-batch on the
        streams' host threads.  Results are identical to the single-stream path."""
        import threading

        main = torch.cuda.current_stream()
        errors = []
        self.step_stats = []
        T, C, h, w = self.T, self.C, self.h, self.w
        host = stack is not None and not stack.is_cuda
        if host:
            # Host ingest: the stack sits in (pinned) host memory; every stream
            # uploads its own sub-batch into a device staging buffer right before it needs it, so the
            # PCIe transfer of one sub-batch overlaps the kernels of the others.
            if self.stage is None:
                self.stage = torch.empty((T, C) + self.tile_grid, dtype=stack.dtype, device=self.dev)
            tiles = self.stage.view(T * C, 1, *self.tile_grid)
        else:
            tiles = stack.view(T * C, 1, *self.tile_grid) if stack is not None else None

        def work(k):
            try:
                stream = self.streams[k]
                stream.wait_stream(main)  # the caller's stream produced the stack / the flat-field pass
                with torch.cuda.stream(stream):
                    for lo, hi in self.ranges[k :: self.n_streams]:
                        if host:
                            self.stage.view(stack.shape)[lo:hi].copy_(stack[lo:hi], non_blocking=True)
                        if tiles is not None:
                            hp.flatfield_stitch(tiles[lo * C : hi * C], self.overlap, flatfield, darkfield, out=self.image[lo:hi],
                                                minmax_out=self.minmax[lo:hi], n_groups=hi - lo)
                        for j, ch in enumerate(self.search_channels):
                            planes = self.image[lo:hi, ch]
                            mm = self.minmax[lo:hi, ch].contiguous()
                            seeds = [(seed + 1000003 * a + 7919 * j) & 0xFFFFFFFFFFFFFFFF for a in range(lo, hi)]
                            res, _ = self.finders[k].find(planes, mm, self.low_q, self.high_q, self.min_roundness,
                                                          self.min_r, seeds)
                            for i, a in enumerate(range(lo, hi)):
                                beads[a] = np.concatenate([beads[a], dedup_against(beads[a], res[i][0], 2 * self.min_r)])
                            f = self.finders[k]
                            self.step_stats.append((torch.stack([f.num_circles.sum(), f.num_alive.sum(), f.num_scored.sum()]),
                                                    int(f.n_edges_host.sum()), dict(f.stats)))
                main.wait_stream(stream)
            except Exception as exc:  # surfaced on the caller's thread
                errors.append(exc)

        threads = [threading.Thread(target=work, args=(k,)) for k in range(self.n_streams)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return beads

    def segment_reduce(self, beads, want_roi=True, image=None, pool_tag=None):
        """fg/bg masks, ROI gather and masked sums for every marker (find.py:561-602)."""
        T, C, h, w = self.T, self.C, self.h, self.w
        image = self.image if image is None else image
        tag = self.pool_tag if pool_tag is None else pool_tag
        # masks straight from the bead tables (mg_roi_segment_reduce): no label map is written or read
        if self.mode == "P":
            return hp.roi_gather_reduce(image.view(T, C, 1, h, w), beads, self.L, None, want_roi=want_roi,
                                        reuse_buffers=True, disks=True, pool_tag=tag)
        # mode R: one assay whose image block is stored (T, C, h, w); gathered in place (time_major)
        return hp.roi_gather_reduce(image.view(1, T, C, h, w), beads, self.L, None, want_roi=want_roi,
                                    reuse_buffers=True, disks=True, time_major=True, pool_tag=tag)

    def __call__(self, stack, flatfield=1.0, darkfield=0.0, seed=0, want_roi=True):
        if self.n_streams > 1 and self.mode == "P":
            beads = [np.empty((0, 3), dtype=np.int32) for _ in range(self.n_assays)]
            beads = self._detect_streams(seed, beads, stack, flatfield, darkfield)  # flat-field per sub-batch
        else:
            if not stack.is_cuda:
                stack = stack.to(self.dev, non_blocking=True)
            tries = self._placement_tries()
            if tries:
                self._trial_image_blocks(stack, flatfield, darkfield, tries[0])
            else:
                self.flatfield(stack, flatfield, darkfield)
            if self.mode == "P" and len(self.search_channels) == 1 and self.batch >= self.n_assays:
                out = self._detect_reduce_on_device(seed, want_roi)
                return self._trial_roi_sets(out, want_roi, tries[1]) if tries else out
            beads = self.detect(seed)
            if tries:
                out = self._trial_pairs(lambda img, tag: self.segment_reduce(beads, want_roi, img, tag), tries[1],
                                        markers=sum(len(b) for b in beads))
                out["beads"] = beads
                return out
        out = self.segment_reduce(beads, want_roi=want_roi)
        out["beads"] = beads
        return out

    # ---- placement trial ------------------------------------------------------------------------------------------
    # The correction pass and the ROI pass each run at one of two levels for the life of a process (3.30 / 3.47 ms,
    # 4.2 / 4.6 ms at 64 x 4 x 4096^2), decided by where the image block and the ROI output set landed in physical
    # memory (not translation, not virtual offsets; between BLOCKS of one process the same two levels
    # show, tools/placement_probe.py).  A process cannot choose where a block lands -- but it can ask for several and keep
    # the best: at the FIRST call of a large processor (either mode) the flat-field passes are timed into a few image blocks and
    # the ROI pass from every one of them into a few output sets; the fastest PAIR stays, the rest is freed before the
    # second call.  Transient memory: tries x the block; time: images x sets ROI passes, once.  MG_PLACEMENT_TRIES="images,sets" (0: off).
    def _placement_tries(self):
        if self._placed:
            return None
        self._placed = True
        try:
            n_img, n_set = (int(v) for v in os.environ.get("MG_PLACEMENT_TRIES", "6,3").split(","))
        except ValueError:
            return None
        block = self.image.numel() * self.image.element_size()
        if (n_img < 2 and n_set < 2) or self.pool_tag or self.n_streams > 1 or block < (1 << 31):
            return None  # (small stacks are bound by latencies; the streaming path alternates two sets of its own)
        free, _ = torch.cuda.mem_get_info(self.dev)
        planes = self.C if self.mode == "P" else self.C * self.T
        roi_set = self.n_assays * 2500 * self.L * self.L * (2 * planes + 2)  # ~ markers x (pixels + masks)
        if free < (max(n_img, 1) + 1) * block + (max(n_set, 1) + 1) * roi_set:
            return None
        self.placement = {}
        return max(n_img, 1), max(n_set, 1)

    def _timed(self, fn, reps=2):
        best = float("inf")
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            b.synchronize()
            best = min(best, a.elapsed_time(b))
        return best

    def _trial_image_blocks(self, stack, flatfield, darkfield, tries):
        """The flat-field passes timed into ``tries`` image blocks.  All of them are kept until the ROI pass has been
        tried from them too (_trial_roi_sets; each holds the corrected stack); the fastest is self.image meanwhile."""
        blocks, times = [], []
        for k in range(tries):
            if k:
                try:
                    self.image = torch.empty_like(self.image)
                except torch.OutOfMemoryError:  # (less free memory than the estimate said: as many blocks as there are)
                    break
            blocks.append(self.image)
            times.append(self._timed(lambda: self.flatfield(stack, flatfield, darkfield)))
        self.image = blocks[int(np.argmin(times))]
        self._trial_blocks = blocks
        self.placement.update(flatfield_ms=[round(t, 3) for t in times])

    def _trial_roi_sets(self, out, want_roi, tries):
        """_trial_pairs for the device-side route: the ROI pass of the call that has just run, from the bead tables where
        the suppression left them."""
        tabs = out.get("device_tables")
        if tabs is None:
            self._trial_blocks = None
            torch.cuda.empty_cache()
            self.placement.update(image_block=int(np.argmin(self.placement["flatfield_ms"])))
            return out
        T, C, h, w = self.T, self.C, self.h, self.w
        counts = [len(b) for b in out["beads"]]
        res = self._trial_pairs(lambda img, tag: hp.roi_gather_reduce(
            img.view(T, C, 1, h, w), None, self.L, None, want_roi=want_roi, reuse_buffers=True, disks=True,
            device_tables=(tabs[0], counts, self.max_r), pool_tag=tag), tries, markers=int(sum(counts)))
        res["beads"] = out["beads"]
        return res

    def _trial_pairs(self, run, tries, markers=None):
        """``run(image block, pool tag)`` = the ROI pass: timed from every image block of the trial into ``tries`` output
        sets of their own (the pass reads one and writes the other: its level belongs to the pair); the pair with the
        smallest flat-field + ROI time is this processor's from now on, everything else goes back to the driver.
        Returns the pass's result in the set that stays.  The sets are as large as the markers make them (a noiseless
        stack has several times the markers of a noisy one): as many are tried as fit the free memory -- fewer than two:
        the image block with the fastest flat-field passes stays and the pass runs once more into the untagged set."""
        blocks, self._trial_blocks = self._trial_blocks, None
        flat_ms = self.placement["flatfield_ms"]
        hp.drop_pool_tags([""])  # (the call's own result is made again below)
        torch.cuda.empty_cache()
        if markers is not None:
            planes = self.C if self.mode == "P" else self.C * self.T
            set_bytes = int(1.3 * markers * self.L * self.L * (2 * planes + 2)) + (64 << 20)
            free, _ = torch.cuda.mem_get_info(self.dev)
            tries = int(min(tries, max(0, free - (8 << 30)) // set_bytes))
        tags = ["#place%d" % k for k in range(tries)]
        roi_ms = None
        if tries >= 2:
            try:
                roi_ms = [[self._timed(lambda: run(img, tag)) for tag in tags] for img in blocks]
            except torch.OutOfMemoryError:  # (the estimate was too low: no set trial)
                roi_ms = None
        if roi_ms is None:
            hp.drop_pool_tags(tags)
            self.image = blocks[int(np.argmin(flat_ms))]
            del blocks
            torch.cuda.empty_cache()
            self.placement.update(image_block=int(np.argmin(flat_ms)), roi_ms=None, roi_set=None)
            return run(self.image, self.pool_tag)
        total = np.asarray(roi_ms) + np.asarray(flat_ms)[:, None]
        bi, bj = (int(v) for v in np.unravel_index(int(np.argmin(total)), total.shape))
        hp.drop_pool_tags([t for k, t in enumerate(tags) if k != bj])
        self.image, self.pool_tag = blocks[bi], tags[bj]
        del blocks
        torch.cuda.empty_cache()  # the blocks not kept go back to the driver NOW (~0.4 s for 50 GB), not at the next graph capture
        self.placement.update(roi_ms=[[round(t, 3) for t in row] for row in roi_ms], image_block=bi, roi_set=bj)
        return run(self.image, self.pool_tag)

    def _detect_reduce_on_device(self, seed, want_roi):
        """One search channel, the whole stack in one batch: there is no cross-channel de-duplication
        (find.py:490-500) to do on the host, so the ROI pass reads the suppression's bead tables where
        they are -- on the device -- and the host fetches its copy of them while that pass runs."""
        T, C, h, w = self.T, self.C, self.h, self.w
        ch = self.search_channels[0]
        seeds = [(seed + 1000003 * a) & 0xFFFFFFFFFFFFFFFF for a in range(self.n_assays)]
        # the ROI pass is queued behind the suppression before the host has seen the bead counts (find's `follow`)
        roi_pass = lambda d_out, d_num, cap: hp.roi_gather_reduce(  # noqa: E731
            self.image.view(T, C, 1, h, w), None, self.L, None, want_roi=want_roi, reuse_buffers=True, disks=True,
            device_tables=(d_out, None, self.max_r), device_counts=(d_num, cap, self._roi_bound), pool_tag=self.pool_tag)
        counts, (d_beads, d_scores, _) = self.finder.find(self.image[:, ch], self.minmax[:, ch], self.low_q,
                                                           self.high_q, self.min_roundness, self.min_r, seeds,
                                                           host_results=False, follow=roi_pass, stable_input=True)
        out = hp.finish_roi(self.finder.follow_result, counts)
        if out is None:  # more markers than the pass was launched for: once more, with the counts
            out = hp.roi_gather_reduce(self.image.view(T, C, 1, h, w), None, self.L, None, want_roi=want_roi,
                                       reuse_buffers=True, disks=True, device_tables=(d_beads, counts, self.max_r),
                                       pool_tag=self.pool_tag)
        # the next call's launch bound: 10 % above this call's markers (its workgroups beyond the real count only cost
        # their start; the per-plane capacity x planes would be ~30 % above)
        self._roi_bound = int(1.1 * int(np.sum(counts))) + 16 * self.n_assays
        out["beads"] = [r[0] for r in self.finder.fetch_results(counts, d_beads, d_scores, overlap=True)]
        return out


def stitched_shape(rows, cols, tile_y, tile_x, overlap):
    """Size of the image Stitcher makes of a (rows, cols) grid of (tile_y, tile_x) tiles (stitch.py:22-39)."""
    clip, rem = overlap // 2, overlap % 2
    return rows * (tile_y - 2 * clip - rem), cols * (tile_x - 2 * clip - rem)


def process_stream(chunks, flatfield=1.0, darkfield=0.0, seed=0, want_roi=False, prefetch=2, overlap=0, sink=None,
                   first_timepoint=None, **processor_kwargs):
    """Mode-P processing of a time series that arrives chunk by chunk (config C5: ``reader.iter_time_chunks``
    or any iterator of (T_chunk, C, H, W) blocks -- or TILED blocks (T_chunk, C, rows, cols, tile_y, tile_x), which the
    flat-field pass crops and joins with ``overlap`` on the device, so the stitched assay never exists on the host --
    optionally wrapped as (time_values, channels, block)).
    A reader thread keeps ``prefetch`` chunks ahead, so decoding files overlaps the GPU's work on the chunk
    before; inside a chunk the upload overlaps compute when ``n_streams > 1`` is passed on.  Every
    timepoint is its own assay, and its RNG stream only depends on its global index: the results equal
    those of the whole stack in one ``StackProcessor`` call, whatever the chunk size.

    ``first_timepoint``: the global index of the first timepoint of ``chunks`` (default: the iterator's own
    ``first_timepoint`` -- ``iter_time_chunks(time_range=...)`` carries it -- else 0).  A rank of a multi-GPU run streams
    its block [lo, hi) of the time axis with ``first_timepoint=lo``: assay indices, seeds (``seed + 1000003 index``) and
    the sink's file names are then those of the unsharded run, and the union of the ranks' results IS that run's result
    (the reference's unit is one assay per loop turn, pipeline.py:18-24, any of which any worker may take).

    Yields one result dict per chunk (as ``StackProcessor.__call__``, plus ``first_timepoint``); ROI pixel
    stacks / masks (``want_roi``) are views of pooled buffers that the chunk after next overwrites (two sets in turn).
    ``sink`` (``magnify_amd.sink.HostSink`` / ``SaveSink``): called with every chunk's result while its buffers are
    valid -- it copies what it keeps to the host beside the next chunk's kernels and turns every timepoint into a
    Dataset with the reference's schema (kept in memory or saved with ``mg.save``): the streamed results persist,
    as the reference's do through ``Dataset.mg.cache`` (accessor.py:18-35, find.py:604).  The sink is closed (its
    pending chunk finished, its writer joined) however the stream ends -- exhausted, abandoned by the consumer or
    broken by an error."""
    import queue
    import threading

    prefetch = max(1, int(prefetch))
    ring = getattr(chunks, "ring", None)
    if ring is not None and ring < prefetch + 2:
        # live blocks: one with the consumer, `prefetch` queued, one the reader holds while the queue is full
        raise ValueError(f"process_stream(prefetch={prefetch}) holds up to {prefetch + 2} blocks, the reader cycles "
                         f"through {ring}: pass ring >= {prefetch + 2} to iter_time_chunks")
    if first_timepoint is None:
        first_timepoint = int(getattr(chunks, "first_timepoint", 0))
    q = queue.Queue(maxsize=prefetch)
    stop, cancelled = object(), threading.Event()

    def produce():
        try:
            for item in chunks:
                while not cancelled.is_set():
                    try:
                        q.put(item, timeout=0.2)
                        break
                    except queue.Full:
                        continue
                if cancelled.is_set():
                    break
            else:
                q.put(stop)
        except BaseException as exc:  # surfaces in the consumer
            q.put(exc)
        finally:
            if cancelled.is_set() and hasattr(chunks, "close"):
                chunks.close()  # (the reader's files and blocks; a generator may only be closed by the thread that runs it)

    producer = threading.Thread(target=produce, daemon=True)
    producer.start()
    # Three Python threads (reader, this one, the sink's writer) share the interpreter lock; with the default switch
    # interval of 5 ms a thread that comes back from a native call can wait that long for a thread busy in bytecode --
    # longer than a whole timepoint takes on the GPU.  A short interval for the duration of the stream.
    import sys

    old_interval = sys.getswitchinterval()
    sys.setswitchinterval(min(old_interval, 2e-4))
    procs, done, n_chunk, copied = {}, int(first_timepoint), 0, [None, None]
    try:
        while True:
            item = q.get()
            if item is stop:
                return
            if isinstance(item, BaseException):
                raise item
            block = item[-1] if isinstance(item, tuple) else item
            if not isinstance(block, torch.Tensor):
                block = torch.from_numpy(np.ascontiguousarray(block))
            key = (tuple(block.shape), block.dtype)
            if key not in procs:  # a shorter last chunk gets its own workspaces
                if block.dim() == 6:  # (T, C, rows, cols, tile_y, tile_x): stitched by the flat-field pass
                    t, c = block.shape[:2]
                    h, w = stitched_shape(*block.shape[2:], overlap)
                    procs[key] = StackProcessor(t, c, h, w, dtype=block.dtype, mode="P", tile_grid=tuple(block.shape[2:]),
                                                overlap=overlap, **processor_kwargs)
                else:
                    t, c, h, w = block.shape
                    procs[key] = StackProcessor(t, c, h, w, dtype=block.dtype, mode="P", **processor_kwargs)
            t = block.shape[0]
            parity = n_chunk & 1
            if copied[parity] is not None:  # the sink's host copy of the chunk that last wrote this buffer set
                torch.cuda.current_stream().wait_event(copied[parity])
            procs[key].pool_tag = f"#{parity}"
            out = procs[key](block, flatfield, darkfield, seed=(seed + 1000003 * done) & 0xFFFFFFFFFFFFFFFF, want_roi=want_roi)
            out["first_timepoint"] = done
            for k in ("sums", "counts"):  # small; the pooled buffers behind them are reused by the next chunk
                if out.get(k) is not None:
                    out[k] = out[k].clone()
            if isinstance(item, tuple) and len(item) == 3:
                out["time"], out["channel"] = item[0], item[1]
            if sink is not None:
                copied[parity] = sink(out)
            done += t
            n_chunk += 1
            yield out
    finally:
        cancelled.set()
        sys.setswitchinterval(old_interval)
        if sink is not None:
            sink.close()


def dedup_against(seen: np.ndarray, new: np.ndarray, radius: float) -> np.ndarray:
    """Cross-channel de-duplication (find.py:490-500): drop new beads that have an earlier bead
    within ``radius`` (Euclidean, inclusive, as KDTree.query_ball_point)."""
    if len(seen) == 0 or len(new) == 0:
        return new
    d2 = ((new[:, None, :2].astype(np.float64) - seen[None, :, :2].astype(np.float64)) ** 2).sum(-1)
    return new[~(d2 <= float(radius) ** 2).any(axis=1)]


# --------------------------------------------------------------------------------------
# synthetic stacks, generated on the device
# --------------------------------------------------------------------------------------


def synthetic_stack(n_t, n_c, h, w, beads_per_mpx=120.0, seed=4000, r_lo=8, r_hi=20, jitter=2, device="cuda",
                    noiseless=False):
    """uint16 stack (T, C, H, W): background 100 + Poisson(20) + N(0, 3) read noise, filled-disk
    beads (the reference's filled_circle_points pixel sets) of radius U{8..20}, per-channel value
    U{500..4000}, non-overlapping, jittered by +-``jitter`` px per timepoint.
    ``noiseless``: zero background and bead value 1000, as in the reference's own tests (both Canny
    quantiles degenerate to 0 there).
    Returns (stack, truth) with truth (n_beads, 3) [row, col, r] of timepoint 0."""
    from . import _native as nat

    rng = np.random.default_rng(seed)
    n_beads = int(round(beads_per_mpx * h * w / 1e6))
    border = r_hi + jitter + 2
    # Jittered-lattice placement: O(n) and non-overlapping by construction.
    min_pitch = 2 * r_hi + 4 + 2 * jitter
    pitch = max(min_pitch, int(math.sqrt((h - 2 * border) * (w - 2 * border) / max(n_beads, 1))))
    while pitch > min_pitch and ((h - 2 * border) // pitch) * ((w - 2 * border) // pitch) < n_beads:
        pitch -= 1
    gy, gx = max((h - 2 * border) // pitch, 0), max((w - 2 * border) // pitch, 0)
    n_beads = min(n_beads, gy * gx)
    cells = rng.choice(gy * gx, size=n_beads, replace=False) if n_beads else np.zeros(0, dtype=np.int64)
    slack = pitch - min_pitch + 1
    rows = border + (cells // max(gx, 1)) * pitch + min_pitch // 2 + rng.integers(0, slack, n_beads)
    cols = border + (cells % max(gx, 1)) * pitch + min_pitch // 2 + rng.integers(0, slack, n_beads)
    radii = rng.integers(r_lo, r_hi + 1, n_beads)
    truth = np.column_stack([rows, cols, radii]).astype(np.int64)
    disks = {}
    for r in range(r_lo, r_hi + 1):
        hw = nat.disk_halfwidths(r)
        pts = [(dy, dx) for dy in range(-r, r + 1) for dx in range(-hw[dy + r], hw[dy + r] + 1)]
        disks[r] = np.asarray(pts, dtype=np.int64)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    stack = torch.empty((n_t, n_c, h, w), dtype=torch.uint16, device=device)
    for t in range(n_t):
        jit = rng.integers(-jitter, jitter + 1, size=(n_beads, 2)) if t > 0 else np.zeros((n_beads, 2), dtype=np.int64)
        flat_idx = np.concatenate([(rows[i] + jit[i, 0] + disks[radii[i]][:, 0]) * w + cols[i] + jit[i, 1] + disks[radii[i]][:, 1]
                                   for i in range(n_beads)]) if n_beads else np.zeros(0, dtype=np.int64)
        owner = np.repeat(np.arange(n_beads), [len(disks[r]) for r in radii]) if n_beads else np.zeros(0, dtype=np.int64)
        d_idx = torch.from_numpy(flat_idx).to(device)
        for c in range(n_c):
            values = rng.integers(500, 4001, n_beads).astype(np.float32)
            if noiseless:
                values[:] = 1000.0
                plane = torch.zeros((h * w,), device=device)
            else:
                plane = 100.0 + torch.poisson(torch.full((h * w,), 20.0, device=device), generator=gen)
                plane += torch.randn((h * w,), device=device, generator=gen) * 3.0
            if n_beads:
                plane.index_add_(0, d_idx, torch.from_numpy(values[owner]).to(device))
            plane = torch.clamp(torch.round(plane), 0, 32767).to(torch.int16)
            stack[t, c] = plane.view(torch.uint16).view(h, w)
    return stack, truth
