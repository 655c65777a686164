import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, "tests")
import numpy as np, torch
from magnify_amd import hotpath as hp
from magnify_amd.stack import synthetic_stack
for T in (1, 8, 64):
    stack, _ = synthetic_stack(T, 1, 4096, 4096, seed=4000)
    cf = hp.CircleFinder(T, 4096, 4096, 5, 25, 1000)
    planes = stack[:, 0]
    for full in (True, False):
        cf.hyst_full = full
        cf._recent_sweeps[:] = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            cf.edge_stage(planes, None, 0.1, 0.9)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        timer = hp.StageTimer(); hp.set_timer(timer)
        cf.edge_stage(planes, None, 0.1, 0.9)
        summ = timer.summary(); hp.set_timer(None)
        msg = f"T={T} full={full} hyst={summ['mg_canny_hysteresis']} sweeps={cf.stats.get('hysteresis_sweeps')}"
        if full:
            c = [int(cf.hyst_dirty.cpu().numpy()[-1]), int(cf.changed[0].sum().item())]
            msg += f" ticket={c[0]} overflow={c[1]} tiles={T*256}"
        print(msg, flush=True)
