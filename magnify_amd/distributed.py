"""Multi-GPU sharding of the hot path: one process per GPU, the time (assay) axis is split in
contiguous blocks, each rank runs the whole per-plane chain locally, and the only exchange is a
variable-length all-gather of the final marker table (plus, in single-assay mode "R", a max
all-reduce of the two flat-field maxima).  Backend "nccl" is RCCL over xGMI on ROCm; the same code
runs on "gloo" for the CPU tests.  The reference has no distributed code.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def _share_gpu(world_local: int) -> bool:
    """Fewer GPUs than ranks on this node (the one-GPU test box): every rank works on cuda:0 and the ranks talk gloo
    (a rehearsal of the multi-rank path, not a measurement).  Decided by every rank for itself -- the launcher's parent
    process does not look at the GPUs at all (magnify_amd/launch.py) -- from the same two numbers, so all ranks agree."""
    if os.environ.get("MG_SHARE_GPU") in ("0", "1"):
        return os.environ["MG_SHARE_GPU"] == "1"
    share = torch.cuda.is_available() and 0 < torch.cuda.device_count() < world_local
    if share:
        os.environ["MG_SHARE_GPU"] = "1"
        os.environ.setdefault("MG_DIST_BACKEND", "gloo")
    return share


def init_from_env(backend=None, single_rank_group=False):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); no-op for 1 rank unless
    ``single_rank_group`` (a process group of one: the collectives below then run through the backend -- RCCL for
    "nccl" -- instead of returning at once)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = _share_gpu(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("MG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = 0 if share else local
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        local = 0 if share else local
        torch.cuda.set_device(local)
    return rank, world, local


def shard_range(n_units: int, rank: int, world: int):
    """Contiguous block of units (timepoints / assays) owned by ``rank``."""
    base, extra = divmod(n_units, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def allreduce_max_(t: torch.Tensor):
    """In-place max all-reduce (flat-field maxima in single-assay mode, preprocess.py:84,86)."""
    if dist.is_initialized():
        if dist.get_backend() == "gloo" and t.is_cuda:
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MAX)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def gather_marker_table(table: torch.Tensor) -> torch.Tensor:
    """All-gather of per-rank marker tables (M_rank, F) -> (sum M_rank, F), rank order.

    Variable length: counts are gathered first, then one padded all-gather."""
    if not dist.is_initialized():
        return table
    world = dist.get_world_size()
    out_device = table.device
    if dist.get_backend() == "gloo":
        table = table.cpu()  # gloo collectives run on host tensors
    count = torch.tensor([table.shape[0]], dtype=torch.int64, device=table.device)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    padded = torch.zeros((cap, table.shape[1]), dtype=table.dtype, device=table.device)
    padded[: table.shape[0]] = table
    parts = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(parts, padded)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0).to(out_device)


def marker_table(out: dict, assay_offset: int, n_channels: int, device) -> torch.Tensor:
    """Pack a StackProcessor result into rows
    [assay, row, col, r, fg_count, bg_count, fg_sum[C], bg_sum[C]] (float64; exact for these ints): one kernel over the
    device-resident bead tables, counts and sums (hotpath.marker_table)."""
    if out.get("sums") is None or out.get("counts") is None:
        raise ValueError("marker_table needs the reductions (want_sums) of the result")
    if out["sums"].is_cuda:
        from . import hotpath

        return hotpath.marker_table(out, assay_offset, n_channels).to(device)
    # results that live in host memory (the gloo rehearsal of the multi-rank path): plain tensor packing
    import numpy as np

    beads = out["beads"]
    m = int(sum(len(b) for b in beads))
    width = 6 + 2 * n_channels
    if m == 0:
        return torch.zeros((0, width), dtype=torch.float64, device=device)
    head = np.concatenate([np.column_stack([np.full(len(b), assay_offset + a), b]) for a, b in enumerate(beads) if len(b)])
    tab = torch.empty((m, width), dtype=torch.float64, device=device)
    tab[:, :4] = torch.from_numpy(head.astype(np.float64)).to(device)
    tab[:, 4:6] = out["counts"].to(torch.float64)
    sums = out["sums"][:, :, 0, :]  # (M, C, 2) at the assay's (only) time index
    tab[:, 6 : 6 + n_channels] = sums[:, :, 0]
    tab[:, 6 + n_channels :] = sums[:, :, 1]
    return tab


def stream_series(pattern, chunk, flatfield=1.0, darkfield=0.0, seed=0, sink=None, rank=None, world=None,
                  pinned=True, workers=None, prefetch=2, on_chunk=None, **stream_kwargs):
    """Config C5 across GPUs (second partition): ONE time series behind ``pattern`` (tiled or not,
    ``reader.iter_time_chunks``' grammar), every rank streams its own contiguous block of the time axis
    (``shard_range``) through its GPU -- files -> page-locked ring -> upload -> stitch + flat-field + detection + ROI
    reduction, ``stack.process_stream(first_timepoint=lo)`` -- and the run ends with the variable-length all-gather of
    the marker table (the only exchange; ROI pixels stay with their rank, which persists them through ``sink``:
    a ``SaveSink`` pattern's ``{index}`` is the GLOBAL timepoint, so the ranks together leave the files of the
    single-process run).  The reference's unit of work is one TIFF page per dask block (reader.py:265-292) and one
    assay per loop turn (pipeline.py:18-24): any worker may take any time range.

    ``rank`` / ``world`` default to the process group's (1 rank without one).  ``workers``: reader threads of THIS rank
    (default: the node's cores divided among its local ranks, at most 16).  ``on_chunk(out)`` sees every chunk's result
    while its pooled buffers are valid.  Returns ``(table, (lo, hi))``: the marker table of the WHOLE series on every
    rank -- rows [timepoint, row, col, r, fg_count, bg_count, fg_sum[C], bg_sum[C]] in time order -- and this rank's
    time range."""
    from . import reader
    from .stack import process_stream

    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    series = pattern if isinstance(pattern, reader.TimeSeries) else reader.TimeSeries(pattern)
    lo, hi = shard_range(len(series), rank, world)
    if workers is None:
        local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
        workers = max(2, min(16, (os.cpu_count() or 1) // local))
    chunks = series.chunks(chunk, time_range=(lo, hi), pinned=pinned and torch.cuda.is_available(), workers=workers,
                           ring=max(4, int(prefetch) + 2) if pinned and torch.cuda.is_available() else None)
    n_c = len(series.channels)
    device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    tables = []
    for out in process_stream(chunks, flatfield, darkfield, seed=seed, sink=sink, prefetch=prefetch, first_timepoint=lo,
                              **stream_kwargs):
        tables.append(marker_table(out, out["first_timepoint"], n_c, device))
        if on_chunk is not None:
            on_chunk(out)
    local_table = torch.cat(tables) if tables else torch.zeros((0, 6 + 2 * n_c), dtype=torch.float64, device=device)
    return gather_marker_table(local_table), (lo, hi)


def broadcast_beads(beads, src: int = 0, device="cpu"):
    """Broadcast a variable-length bead table (M, 3) int32 from rank ``src`` (collective 2:
    in single-assay mode the beads found at time 0 serve every time shard).  ``beads`` is ignored on the
    other ranks.  Returns a numpy (M, 3) int32 array on every rank."""
    import numpy as np

    if not dist.is_initialized():
        return np.asarray(beads, dtype=np.int32).reshape(-1, 3)
    dev = torch.device("cpu") if dist.get_backend() == "gloo" else torch.device(device)
    me = dist.get_rank()
    count = torch.tensor([len(beads) if me == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(count, src)
    table = torch.zeros((int(count.item()), 3), dtype=torch.int32, device=dev)
    if me == src and len(beads):
        table.copy_(torch.from_numpy(np.ascontiguousarray(beads, dtype=np.int32).reshape(-1, 3)))
    if table.numel():
        dist.broadcast(table, src)
    return table.cpu().numpy()


def run_mode_r(proc, stack_local: torch.Tensor, flatfield=1.0, darkfield=0.0, seed=0, want_roi=True):
    """One single-assay (mode "R") step with the TIME axis sharded over the ranks:

    1. the flat-field maxima span the whole array (preprocess.py:84,86): local maxima, then a max
       all-reduce of the two doubles;
    2. detection happens on global time 0 only (find.py:477): the rank that owns it (rank 0 under
       ``shard_range``) detects and broadcasts the bead table;
    3. every rank gathers and reduces the windows of its own timepoints with those beads.

    ``proc`` is a ``StackProcessor(mode="R")`` sized for the local shard ``stack_local (T_local, C, H, W)``.
    Returns the local result dict (``roi`` is (M, C, T_local, L, L)); identical, shard for shard, to the
    single-process result on the whole stack."""
    from . import hotpath as hp

    if proc.mode != "R":
        raise ValueError("run_mode_r needs a StackProcessor(mode='R')")
    t, c, h, w = stack_local.shape
    tiles = stack_local.view(t * c, 1, 1, 1, h, w)
    max2 = allreduce_max_(hp.flatfield_max(tiles, flatfield, darkfield, 1))
    proc.flatfield(stack_local, flatfield, darkfield, max2=max2)
    owner = 0
    me = dist.get_rank() if dist.is_initialized() else 0
    beads = proc.detect(seed)[0] if me == owner else None
    beads = broadcast_beads(beads, owner, device=stack_local.device)
    out = proc.segment_reduce([beads], want_roi=want_roi)
    out["beads"] = [beads]
    return out
