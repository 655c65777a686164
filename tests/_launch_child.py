"""Child of tests/test_cpu_distributed.py::test_self_launcher: one rank started by magnify_amd.launch.spawn_ranks.
Gathers a marker table over gloo and prints one JSON line on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from magnify_amd import distributed as mgd  # noqa: E402

rank, world, local = mgd.init_from_env(backend="gloo")
lo, hi = mgd.shard_range(int(sys.argv[1]), rank, world)
table = torch.full((hi - lo, 3), float(rank), dtype=torch.float64)
table[:, 1] = torch.arange(lo, hi, dtype=torch.float64)
full = mgd.gather_marker_table(table)
if len(sys.argv) > 2 and int(sys.argv[2]) == rank:
    raise SystemExit(7)  # a failing rank must take the whole launch down
if rank == 0:
    print(json.dumps({"world": world, "rows": full[:, 1].tolist(), "owners": full[:, 0].tolist(),
                      "launched": os.environ.get("MG_LAUNCHED")}))
else:
    print(f"rank {rank} done")  # must not reach the parent's stdout
torch.distributed.barrier()
torch.distributed.destroy_process_group()
