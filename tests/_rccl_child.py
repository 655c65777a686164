"""Child of tests/test_gpu_distributed.py::test_rccl_collectives_on_one_rank: a process group of ONE rank on backend
"nccl" (= RCCL on ROCm), started before anything else touches the GPU; the marker-table all-gather, the max
all-reduce and the bead broadcast go through their device-tensor branches (the branches an 8-GPU run takes).
Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from magnify_amd import distributed as mgd  # noqa: E402
from magnify_amd.stack import StackProcessor, synthetic_stack  # noqa: E402

rank, world, local = mgd.init_from_env(backend="nccl", single_rank_group=True)
dev = torch.device("cuda", local)
backend = torch.distributed.get_backend()
beads = [np.array([[10, 20, 5], [30, 40, 6]], dtype=np.int32), np.empty((0, 3), dtype=np.int32), np.array([[7, 8, 9]], dtype=np.int32)]
out = {"beads": beads, "counts": torch.arange(6, dtype=torch.int32, device=dev).reshape(3, 2),
       "sums": torch.arange(3 * 2 * 2, dtype=torch.float64, device=dev).reshape(3, 2, 1, 2)}
local_table = mgd.marker_table(out, 5, 2, dev)
table = mgd.gather_marker_table(local_table)
mx = mgd.allreduce_max_(torch.tensor([3.5, -1.0], dtype=torch.float64, device=dev))
shared = mgd.broadcast_beads(np.array([[1, 2, 3], [4, 5, 6]]), src=0, device=dev)
empty = mgd.broadcast_beads(np.empty((0, 3), np.int32), src=0, device=dev)
# mode R with its flat-field max all-reduce and bead broadcast through RCCL
stack = synthetic_stack(2, 2, 256, 320, seed=5, beads_per_mpx=300.0)[0]
proc = StackProcessor(2, 2, 256, 320, num_iter=40000, search_channels=(0,), mode="R")
got = mgd.run_mode_r(proc, stack, 0.9, 100.0, seed=3)
want = StackProcessor(2, 2, 256, 320, num_iter=40000, search_channels=(0,), mode="R")(stack, 0.9, 100.0, seed=3)
torch.distributed.barrier()
print(json.dumps({"backend": backend, "world": world, "table_on_device": bool(table.is_cuda),
                  "table_equal": bool(torch.equal(table, local_table)), "max": mx.tolist(), "shared": shared.tolist(),
                  "empty_shape": list(empty.shape), "mode_r_beads": int(len(got["beads"][0])),
                  "mode_r_equal": bool(np.array_equal(got["beads"][0], want["beads"][0]) and torch.equal(got["sums"], want["sums"]))}))
torch.distributed.destroy_process_group()
