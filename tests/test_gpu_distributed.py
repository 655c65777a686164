"""Two ranks on the one GPU of the test box (gloo between them, MG_SHARE_GPU rehearsal of the
multi-GPU path): time-sharded single-assay mode (flat-field max all-reduce + bead-table broadcast) and
the weak-scaling marker-table gather against the single-process results."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T, C, H, W = 4, 2, 384, 512


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _stack():
    from magnify_amd.stack import synthetic_stack

    return synthetic_stack(T, C, H, W, seed=321, beads_per_mpx=300.0)[0]


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), MG_SHARE_GPU="1", MG_DIST_BACKEND="gloo")
    import torch

    from magnify_amd import distributed as mgd
    from magnify_amd.stack import StackProcessor

    r, w, _ = mgd.init_from_env()
    stack = _stack()  # same seed on every rank: the full stack, of which a rank uses its shard
    lo, hi = mgd.shard_range(T, r, w)
    proc = StackProcessor(hi - lo, C, H, W, num_iter=60000, search_channels=(0,), mode="R")
    out = mgd.run_mode_r(proc, stack[lo:hi].contiguous(), 0.9, 100.0, seed=4)
    ret[rank] = (out["beads"][0].copy(), out["roi"].cpu().numpy().copy(), out["sums"].cpu().numpy().copy(),
                 proc.image.cpu().numpy().copy(), (lo, hi))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_mode_r_time_sharded_matches_single_process():
    import torch.multiprocessing as mp

    from magnify_amd import hotpath
    from magnify_amd.stack import StackProcessor

    hotpath.require_gpu()
    stack = _stack()
    proc = StackProcessor(T, C, H, W, num_iter=60000, search_channels=(0,), mode="R")
    want = proc(stack, 0.9, 100.0, seed=4)
    assert len(want["beads"][0]) >= 20
    world, port = 2, _free_port()
    manager = mp.get_context("spawn").Manager()
    ret = manager.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for rank in range(world):
        beads, roi, sums, image, (lo, hi) = ret[rank]
        np.testing.assert_array_equal(beads, want["beads"][0])  # broadcast from the owner of time 0
        np.testing.assert_array_equal(image, proc.image[lo:hi].cpu().numpy())  # global maxima on every shard
        np.testing.assert_array_equal(roi, want["roi"][:, :, lo:hi].cpu().numpy())
        np.testing.assert_array_equal(sums, want["sums"][:, :, lo:hi].cpu().numpy())
