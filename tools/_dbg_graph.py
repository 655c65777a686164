import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, "tests")
import numpy as np, torch
import magnify_amd as mg
from magnify_amd import hotpath as hp, find
from synth import draw_beads
mg.seed(1234)
poss = [[[200, 200], [200, 800], [512, 512], [800, 200], [800, 800]], [[50, 512], [974, 512], [512, 50], [512, 974]], [[500, 500], [500, 530]],
        [[300, 300], [600, 600]], [[100, 100]]]
for pos in poss:
    xp = mg.beads(data=mg.DataArray(data=draw_beads((1024, 1024), pos), dims=("y", "x")), min_bead_diameter=16, max_bead_diameter=24,
                  overlap=0, num_iter=10000)
    cf = list(find._FINDERS.values())[-1]
    st = cf.status_host.numpy()
    print(len(pos), "->", xp.roi.sizes["mark"], cf.calls, "replays", cf.graph_replays, "captures", cf.graph_captures, cf.stats.get("graph_error"),
          "edges", st[0], "unresolved", st[2], "alive", st[3], "out", st[4], "scored", st[5], "surv", st[6], "circles", st[7],
          "changed", st[8:14, 0], "undecided", st[8 + 64: 8 + 70, 0], flush=True)
