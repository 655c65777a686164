"""Where the results of a streamed run go.

``stack.process_stream`` hands out, chunk by chunk, views of pooled device buffers that the chunk after next
overwrites; the reference keeps every assay's ``roi / fg / bg`` by spilling them to a zarr store
(``Dataset.mg.cache``, accessor.py:18-35, called at find.py:604) and returns one Dataset per assay
(pipeline.py:14-29).  A sink is what does that here: ``process_stream(..., sink=s)`` calls ``s(out)`` for every chunk
while its buffers are valid; the sink copies what it keeps to page-locked host memory on a side stream (the copy of
chunk k runs beside the kernels of chunk k + 1: the pooled buffers alternate between two sets) and builds, per
timepoint (= assay, mode P), a Dataset with the reference's schema (find.py:503-555):
``roi (mark, channel, time, roi_y, roi_x)``, coords ``fg, bg (mark, time, roi_y, roi_x)``, ``x, y, valid (mark, time)``
(+ ``time`` / ``channel`` labels when the reader gave them).

  * ``HostSink``: keeps the assays in host memory (``.assays``), optionally without ROI pixels.
  * ``SaveSink(pattern)``: writes every assay with ``mg.save`` as soon as its copy has landed (``pattern`` takes
    ``{index}`` = GLOBAL timepoint index: the ranks of a multi-GPU run, each streaming its block of the time axis with
    ``process_stream(first_timepoint=lo)``, share one pattern and leave exactly the files of the single-process run)
    and keeps only the file names: nothing accumulates, a terabyte-scale series leaves a directory of NetCDF files that
    ``mg.load`` reads.

The copies land in a ring of page-locked staging blocks and a WRITER THREAD turns them into Datasets / files: the thread
that feeds the GPU queues the copy and goes on (it waits only when ``depth`` chunks are already on their way).
"""
from __future__ import annotations

import queue
import threading
import time

import numpy as np
import torch

from . import file as mgfile
from .xr_lite import DataArray, Dataset


class _Staging:
    """Page-locked byte blocks for the device-to-host copies of the sink, reused from chunk to chunk (pinning a fresh
    block per chunk and key costs ~0.15 s per GB -- more than the copy).  A block is handed out at least as large as
    asked for (a chunk's marker count varies) and comes back when the writer thread is done with the chunk."""

    def __init__(self):
        self._free, self._lock = [], threading.Lock()

    def take(self, nbytes):
        with self._lock:
            fit = [i for i, b in enumerate(self._free) if b.numel() >= nbytes]
            if fit:
                return self._free.pop(min(fit, key=lambda i: self._free[i].numel()))
            if self._free:  # the largest free block is too small: it makes room for a larger one
                self._free.pop(max(range(len(self._free)), key=lambda i: self._free[i].numel()))
        return torch.empty((max(int(nbytes * 1.25), 1 << 20),), dtype=torch.uint8).pin_memory()

    def give(self, blocks):
        with self._lock:
            self._free.extend(blocks)


_STAGING = _Staging()  # kept by the module, like the reader's blocks: a second stream (or sink) pins nothing new


def release_staging():
    """Give the sinks' page-locked staging blocks back."""
    global _STAGING
    _STAGING = _Staging()


class _Sink:
    """``depth``: how many chunks may be on their way (copy in flight / being written) before ``__call__`` waits for
    the writer -- the GPU feed never waits for a file while the writer keeps up."""

    copy_out = False    # True: the assays outlive the staging blocks (HostSink keeps them): copied out of them
    take_threads = 1    # assays of one chunk handed to `take` side by side (SaveSink: one file each)
    _pool = None
    big_endian = False  # True: multi-byte values are byte-swapped ON THE DEVICE before they travel (SaveSink: NetCDF-3 is
                        # big-endian -- the writer then puts the staging block's bytes into the file as they lie)

    def __init__(self, want_roi=True, want_masks=True, depth=3):
        self.want_roi, self.want_masks = want_roi, want_masks
        self._side = None
        self._staging = _STAGING
        self._slots = threading.Semaphore(max(1, int(depth)))
        self._jobs = queue.Queue()
        self._writer = None
        self._error = None
        # where the writer thread's time went (seconds): waiting for the device-to-host copy, building the assays'
        # Datasets, `take` (HostSink: storing, SaveSink: mg.save); and how long __call__ waited for a free slot
        self.stats = {"chunks": 0, "copy_wait_s": 0.0, "build_s": 0.0, "take_s": 0.0, "slot_wait_s": 0.0, "bytes": 0}

    # -- called by process_stream ---------------------------------------------------------------------------------
    def __call__(self, out):
        """Queue the device-to-host copy of this chunk's outputs (side stream, page-locked staging blocks) and hand the
        chunk to the writer thread, which waits for the copy, builds the assays' Datasets and ``take``s them."""
        self._raise_writer_error()
        dev_keys = [k for k in ("sums", "counts") + (("roi",) if self.want_roi else ()) + (("fg", "bg") if self.want_masks else ())
                    if out.get(k) is not None]
        t0 = time.perf_counter()
        self._slots.acquire()
        self.stats["slot_wait_s"] += time.perf_counter() - t0
        if self._writer is None:
            self._writer = threading.Thread(target=self._write_loop, daemon=True)
            self._writer.start()
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        ready = torch.cuda.Event()
        ready.record(main)
        host, blocks, swapped = {}, [], set()
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            for k in dev_keys:
                src = out[k]
                if src.numel():
                    size = src.element_size()
                    block = self._staging.take(src.numel() * size)
                    blocks.append(block)
                    dst = block[: src.numel() * size].view(src.dtype).view(src.shape)
                    src.record_stream(self._side)
                    if self.big_endian and size > 1:  # (one gather kernel on the side stream, beside the next chunk's work)
                        src = src.contiguous().view(torch.uint8).view(-1, size).flip(1).view(src.dtype).view(src.shape)
                        swapped.add(k)
                    dst.copy_(src, non_blocking=True)
                else:
                    dst = torch.empty(src.shape, dtype=src.dtype)
                host[k] = dst
            done = torch.cuda.Event()
            done.record(self._side)
        meta = {"beads": [np.asarray(b) for b in out["beads"]], "offsets": np.asarray(out["offsets"]), "swapped": swapped,
                "first": int(out.get("first_timepoint", 0)), "time": out.get("time"), "channel": out.get("channel")}
        self._jobs.put((done, host, meta, blocks))
        return done  # process_stream makes the chunk after next wait for it (that one reuses these device buffers)

    def close(self):
        """Finish what is on its way (also when the stream broke off) and stop the writer; re-raises its error."""
        if self._writer is not None:
            self._jobs.put(None)
            self._writer.join()
            self._writer = None
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None
        self._raise_writer_error()

    def _raise_writer_error(self):
        if self._error is not None:
            err, self._error = self._error, None
            raise err

    def _write_loop(self):
        while True:
            job = self._jobs.get()
            if job is None:
                return
            done, host, meta, blocks = job
            try:
                if self._error is None:  # (after an error the remaining chunks are only released)
                    self._finish(done, host, meta)
            except BaseException as exc:
                self._error = exc
            finally:
                done.synchronize()  # (a block must not go back while its copy is in flight)
                self._staging.give(blocks)
                self._slots.release()

    # -- per assay ---------------------------------------------------------------------------------------------------
    def _finish(self, done, host, meta):
        t0 = time.perf_counter()
        done.synchronize()
        t1 = time.perf_counter()
        arrays = {k: (np.array(v.numpy()) if self.copy_out else v.numpy()) for k, v in host.items()}
        for k in meta["swapped"]:  # the block's bytes are big-endian: say so in the dtype (the values read correctly)
            arrays[k] = arrays[k].view(arrays[k].dtype.newbyteorder(">"))
        off = meta["offsets"]
        st = self.stats
        st["chunks"] += 1
        st["copy_wait_s"] += t1 - t0
        st["bytes"] += sum(v.nbytes for v in arrays.values())
        st["build_s"] += time.perf_counter() - t1
        t2 = time.perf_counter()
        assays = []
        for a, beads in enumerate(meta["beads"]):
            lo, hi = int(off[a]), int(off[a + 1])
            assays.append((meta["first"] + a, self.assay_dataset(beads, {k: v[lo:hi] for k, v in arrays.items()},
                                                                 None if meta["time"] is None else meta["time"][a], meta["channel"])))
        t3 = time.perf_counter()
        if self.take_threads > 1 and len(assays) > 1:
            # the assays of a chunk are independent files: written side by side (writes into ONE file queue up behind its
            # inode lock -- 2.6 GB/s whatever the thread count --, writes into different files do not)
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor

                self._pool = ThreadPoolExecutor(max_workers=self.take_threads)
            for job in [self._pool.submit(self.take, index, ds) for index, ds in assays]:
                job.result()
        else:
            for index, ds in assays:
                self.take(index, ds)
        st["build_s"] += t3 - t2
        st["take_s"] += time.perf_counter() - t3

    @staticmethod
    def assay_dataset(beads, arrays, time_label=None, channels=None):
        """One timepoint's results as the reference's Dataset (find.py:503-555, after BeadFinder)."""
        m = len(beads)
        xy = np.asarray(beads, dtype=np.float64).reshape(m, 3)  # [row, col, r]
        coords = {"x": (("mark", "time"), xy[:, 1:2].copy()), "y": (("mark", "time"), xy[:, 0:1].copy()),
                  "valid": (("mark", "time"), np.ones((m, 1), dtype=bool))}
        for k in ("fg", "bg"):
            if k in arrays:
                coords[k] = (("mark", "time", "roi_y", "roi_x"), arrays[k].astype(bool)[:, None])
        ds = Dataset()
        if "roi" in arrays:
            from .utils import roi_mark_chunk

            n_c, length = arrays["roi"].shape[1], arrays["roi"].shape[-1]
            per = {"mark": roi_mark_chunk(m, n_c, 1, length)}  # the reference's chunk policy (find.py:506-531)
            ds["roi"] = DataArray(arrays["roi"], ("mark", "channel", "time", "roi_y", "roi_x")).chunk(per)
        # the fused reductions (not part of the reference's schema; what its users compute from roi / fg / bg)
        ds["fg_sum"] = DataArray(arrays["sums"][..., 0], ("mark", "channel", "time"))
        ds["bg_sum"] = DataArray(arrays["sums"][..., 1], ("mark", "channel", "time"))
        ds["fg_count"] = DataArray(arrays["counts"][:, 0].astype(np.int32), ("mark",))
        ds["bg_count"] = DataArray(arrays["counts"][:, 1].astype(np.int32), ("mark",))
        ds["radius"] = DataArray(xy[:, 2].astype(np.int32), ("mark",))
        ds = ds.assign_coords(coords)
        if time_label is not None:
            ds = ds.assign_coords(time=(("time",), np.asarray([time_label])))
        if channels is not None and "roi" in arrays and len(channels) == arrays["roi"].shape[1]:
            ds = ds.assign_coords(channel=(("channel",), np.asarray(channels)))
        return ds

    def take(self, index, ds):
        raise NotImplementedError


class HostSink(_Sink):
    """Every assay (timepoint) of the stream as a Dataset in host memory: ``sink.assays[index]``."""

    copy_out = True

    def __init__(self, want_roi=True, want_masks=True, depth=3):
        super().__init__(want_roi, want_masks, depth)
        self.assays = {}

    def take(self, index, ds):
        self.assays[index] = ds


class SaveSink(_Sink):
    """Every assay written with ``mg.save`` to ``pattern.format(index=...)`` as soon as it is on the host."""

    big_endian = True

    def __init__(self, pattern, want_roi=True, want_masks=True, shard_bytes=None, depth=3, writers=8):
        super().__init__(want_roi, want_masks, depth)
        self.take_threads = max(1, int(writers))
        if "{index" not in pattern:
            raise ValueError("SaveSink: the file pattern needs an {index} field")
        self.pattern, self.shard_bytes, self.files = pattern, shard_bytes, {}

    def take(self, index, ds):
        path = self.pattern.format(index=index)
        mgfile.save(path, ds, shard_bytes=self.shard_bytes, threads=1 if self.take_threads > 1 else None)
        self.files[index] = path
