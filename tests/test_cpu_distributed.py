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
