import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
import magnify_amd as mg
from magnify_amd import find as mgfind
from synth import draw_chip
KW = dict(min_button_diameter=16, max_button_diameter=32, overlap=0, row_dist=100, col_dist=100)
mg.seed(4321)
for noise, d1 in ((True, 28), (True, 20), (False, 28), (True, 24)):
    a = draw_chip((3, 3), 20); b = draw_chip((3, 3), d1, offset=(10, 10)); a[a > 0] = 3000
    rng = np.random.default_rng(8)
    data = np.stack([a, b]).astype(np.int64)
    if noise: data = data + rng.integers(90, 120, size=data.shape)
    data = data.astype(np.uint16)
    orig = mgfind.ButtonFinder.find_centers
    def spy(self, planes, tag, seeds):
        try:
            gx, gy = orig(self, planes, tag, seeds)
            print("   centres x", np.round(gx, 1).tolist(), "y", np.round(gy, 1).tolist())
            return gx, gy
        except Exception as e:
            print("   find_centers raised", type(e).__name__, e); raise
    mgfind.ButtonFinder.find_centers = spy
    try:
        pipe = mg.microfluidic_chip_pipe(shape=(3, 3), num_iter=5000, search_timestep=[0, 1], **KW)
        pipe.remove_pipe("restore_format")
        xp = pipe(mg.DataArray(data=data, dims=("time", "y", "x"), coords={"time": [0, 1]}))
        print(noise, d1, "ok", xp.sizes)
    except Exception as e:
        print(noise, d1, "raised", type(e).__name__, e)
    mgfind.ButtonFinder.find_centers = orig
