"""Does a hipGraph of the optimistic chain buy anything?  Eager launches vs. graph replay, P = 1 and 8 planes."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, "tests")
import numpy as np, torch
from magnify_amd import hotpath as hp
from magnify_amd.stack import synthetic_stack

for P in (1, 8):
    stack, _ = synthetic_stack(P, 1, 4096, 4096, seed=4000)
    planes = stack[:, 0].contiguous()
    mm = hp.plane_minmax(planes)
    cf = hp.CircleFinder(P, 4096, 4096, 5, 25, 5_000_000)
    seeds = list(range(P))
    for _ in range(3):
        cf.find(planes, mm, 0.1, 0.9, 0.3, 5, seeds, host_results=False)
    torch.cuda.synchronize()

    def chain():
        cf.status.zero_()
        cf.edge_stage(planes, mm, 0.1, 0.9, optimistic=True)
        cf.circle_stage(seeds, 0.3, dedup_centres=True, counters_clear=True)
        return cf.nms_stage(5, optimistic=True)

    def timed(fn, n=30):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n):
            fn(); cf._fetch_status()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def eager():
        bufs, rounds = chain()
        return bufs

    t_eager = timed(eager)
    # GPU-only time of the eager chain (events)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); eager(); e1.record(); torch.cuda.synchronize()
    gpu_eager = e0.elapsed_time(e1)
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(side):
            chain()  # warm-up on the side stream
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                bufs, rounds = chain()
        torch.cuda.synchronize()
        t_graph = timed(lambda: g.replay())
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        gpu_graph = e0.elapsed_time(e1)
        st = cf._fetch_status()
        print(f"P={P}: eager {t_eager:.3f} ms/call (GPU {gpu_eager:.3f}), graph {t_graph:.3f} ms/call (GPU {gpu_graph:.3f}); "
              f"beads {st[4].tolist()[:4]}", flush=True)
    except Exception as exc:
        print(f"P={P}: eager {t_eager:.3f} ms/call (GPU {gpu_eager:.3f}); graph capture failed: {type(exc).__name__}: {exc}", flush=True)
