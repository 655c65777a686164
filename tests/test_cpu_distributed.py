"""World-size-2 gloo test of the multi-GPU plumbing (time-axis sharding + variable-length marker
table all-gather + flat-field max all-reduce), on CPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from magnify_amd import distributed as mgd

    r, w, _ = mgd.init_from_env(backend="gloo")
    lo, hi = mgd.shard_range(7, r, w)
    # each rank "finds" a different number of markers per owned timepoint
    beads = [np.column_stack([np.full(t + 1, 10 * t), np.arange(t + 1), np.full(t + 1, 5)]).astype(np.int32)
             for t in range(lo, hi)]
    m = sum(len(b) for b in beads)
    out = {"beads": beads, "counts": torch.arange(2 * m, dtype=torch.int32).reshape(m, 2),
           "sums": torch.arange(m * 3 * 2, dtype=torch.float64).reshape(m, 3, 1, 2)}
    table = mgd.gather_marker_table(mgd.marker_table(out, lo, 3, torch.device("cpu")))
    mx = mgd.allreduce_max_(torch.tensor([float(r), 10.0 - r], dtype=torch.float64))
    shared = mgd.broadcast_beads(np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]]) if r == 1 else None, src=1)
    empty = mgd.broadcast_beads(np.empty((0, 3), np.int32) if r == 0 else None, src=0)
    assert shared.tolist() == [[1, 2, 3], [4, 5, 6], [7, 8, 9]] and empty.shape == (0, 3)
    ret[rank] = (table.numpy().copy(), mx.numpy().copy(), (lo, hi))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_gloo_marker_table_gather():
    world = 2
    port = _free_port()
    manager = mp.get_context("spawn").Manager()
    ret = manager.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    t0, m0, r0 = ret[0]
    t1, m1, r1 = ret[1]
    np.testing.assert_array_equal(t0, t1)  # every rank holds the same full table
    assert r0 == (0, 4) and r1 == (4, 7)
    assert t0.shape == (sum(t + 1 for t in range(7)), 6 + 2 * 3)
    assert t0[:, 0].tolist() == sorted(t0[:, 0].tolist())  # rank order == time order
    assert set(t0[:, 0].astype(int).tolist()) == set(range(7))
    np.testing.assert_array_equal(m0, [1.0, 10.0])
    np.testing.assert_array_equal(m0, m1)


def test_shard_range_covers_everything():
    from magnify_amd.distributed import shard_range

    for n in (1, 7, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_self_launcher(capsys):
    """bench.py's way of starting N ranks from a parent that never touches the GPU
    (magnify_amd/launch.py): env as torchrun sets it, rank 0's stdout relayed, failures propagated."""
    import io
    import json

    from magnify_amd import launch

    child = os.path.join(ROOT, "tests", "_launch_child.py")
    buf = io.StringIO()
    assert launch.spawn_ranks([child, "7"], 2, share_gpu=False, timeout=120, out=buf) == 0
    lines = [ln for ln in buf.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and "rank 1 done" not in buf.getvalue()
    rec = json.loads(lines[0])
    assert rec == {"world": 2, "rows": [float(i) for i in range(7)], "owners": [0.0] * 4 + [1.0] * 3, "launched": "1"}
    # a rank that exits non-zero fails the launch (the other rank is terminated, not left behind)
    assert launch.spawn_ranks([child, "7", "1"], 2, share_gpu=False, timeout=120, out=io.StringIO()) == 7
    env = launch.rank_env(1, 4, 1234, share_gpu=True, base={})
    assert env["RANK"] == "1" and env["WORLD_SIZE"] == "4" and env["MG_SHARE_GPU"] == "1" and env["MG_DIST_BACKEND"] == "gloo"
    assert launch.launched_by_torchrun({"RANK": "0", "WORLD_SIZE": "2"}) and not launch.launched_by_torchrun({})


def _write_tiled_series(root, n_t, n_c, rows, cols, ty, tx, seed=3):
    """config C5's file shape: one OME-BigTIFF per tile position holding (time, channel) pages."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from tiffwrite import ome_xml, write_tiff

    rng = np.random.default_rng(seed)
    data = rng.integers(0, 65535, (n_t, n_c, rows, cols, ty, tx)).astype(np.uint16)
    for r in range(rows):
        for c in range(cols):
            pages = [data[t, ch, r, c] for t in range(n_t) for ch in range(n_c)]
            write_tiff(os.path.join(root, f"acq_r{r}_c{c}.ome.tif"), pages, bigtiff=True,
                       description=ome_xml(size_c=n_c, size_t=n_t, size_y=ty, size_x=tx, channel_names=["a", "b"][:n_c]))
    return data, os.path.join(root, "acq_r(row)_c(col).ome.tif")


def _stream_worker(rank, world, port, pattern, chunk, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from magnify_amd import distributed as mgd
    from magnify_amd import reader

    r, w, _ = mgd.init_from_env(backend="gloo")
    it = reader.iter_time_chunks(pattern, chunk, rank=r, world=w, workers=2)
    rows, blocks, t = [], [], it.first_timepoint
    for stamps, channels, block in it:
        for i in range(block.shape[0]):  # one "marker" row per timepoint: [global timepoint, checksum, rank]
            rows.append([t, float(block[i].astype(np.uint64).sum()), r])
            t += 1
        blocks.append(block.copy())
    table = mgd.gather_marker_table(torch.tensor(rows, dtype=torch.float64).reshape(-1, 3))
    ret[rank] = (table.numpy().copy(), np.concatenate(blocks), it.first_timepoint, (it.lo, it.hi))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_gloo_streamed_series_partition(tmp_path):
    """Config C5 across ranks: two gloo ranks stream the two halves of ONE tiled OME-BigTIFF series
    (reader.iter_time_chunks(rank=, world=)); the blocks they read are the series' own timepoints, numbered globally,
    and the gathered table equals the single-process stream's, row for row."""
    from magnify_amd import reader

    n_t, chunk = 7, 2
    data, pattern = _write_tiled_series(str(tmp_path), n_t, 2, 2, 3, 24, 40)
    world, port = 2, _free_port()
    manager = mp.get_context("spawn").Manager()
    ret = manager.dict()
    mp.spawn(_stream_worker, args=(world, port, pattern, chunk, ret), nprocs=world, join=True)
    whole = np.concatenate([b for _, _, b in reader.iter_time_chunks(pattern, chunk)])
    np.testing.assert_array_equal(whole, data)
    want = np.array([[t, float(data[t].astype(np.uint64).sum())] for t in range(n_t)])
    for rank in range(world):
        table, blocks, first, (lo, hi) = ret[rank]
        assert (lo, hi) == ((0, 4), (4, 7))[rank] and first == lo
        np.testing.assert_array_equal(blocks, data[lo:hi])
        np.testing.assert_array_equal(table[:, :2], want)  # every rank holds the whole series' table, in time order
        assert table[:, 2].tolist() == [0.0] * 4 + [1.0] * 3
    # an explicit range, and the refusals
    it = reader.iter_time_chunks(pattern, 3, time_range=(2, 6))
    got = list(it)
    assert it.first_timepoint == 2 and [len(g[0]) for g in got] == [3, 1] and got[0][0] == [2, 3, 4]
    np.testing.assert_array_equal(np.concatenate([g[2] for g in got]), data[2:6])
    assert list(reader.iter_time_chunks(pattern, 3, time_range=(5, 5))) == []
    with pytest.raises(ValueError):
        reader.iter_time_chunks(pattern, 3, time_range=(2, 9))
    with pytest.raises(ValueError):
        reader.iter_time_chunks(pattern, 3, time_range=(0, 2), rank=0, world=2)
