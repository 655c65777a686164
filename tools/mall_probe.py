"""Does the 256 MB Infinity Cache keep one assay (4 x 32 MB + the 64 MB flat image) between the maxima pass and the
correction pass?  Times the correction of an assay right after its own maxima pass (warm) and after another assay's (cold)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ".")); sys.path.insert(0, "tests")
import numpy as np, torch
from magnify_amd import hotpath as hp
from magnify_amd.stack import synthetic_stack
from synth import vignette

S, C, A = 4096, 4, 8
stack, _ = synthetic_stack(A, C, S, S, seed=4000)
flat = torch.from_numpy(vignette((S, S))).cuda()
out = torch.empty_like(stack)
mm = torch.empty((A * C, 2), dtype=torch.float64, device="cuda")
tiles = stack.view(A * C, 1, 1, 1, S, S)
def fmax(a):
    return hp.flatfield_max(tiles[a * C:(a + 1) * C], flat, 100.0, 1)
def fapply(a, m2):
    hp.flatfield_stitch(tiles[a * C:(a + 1) * C], 0, flat, 100.0, max2=m2, out=out[a], minmax_out=mm[a * C:(a + 1) * C], n_groups=1)
m2s = [fmax(a) for a in range(A)]
torch.cuda.synchronize()
for mode, shift in (("warm", 0), ("cold", 4), ("warm", 0), ("cold", 4)):
    ev = []
    for it in range(40):
        a = it % A
        fmax(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fapply((a + shift) % A, m2s[(a + shift) % A]); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev][8:])
    print(f"{mode}: correction of one assay (128 MiB in, 128 MiB out, 64 MiB flat) {np.median(t)*1e3:.1f} us median, {t.min()*1e3:.1f} min "
          f"-> {(256 + 64) * 1.048576 / np.median(t):.0f} GB/s", flush=True)
# maxima pass alone, warm (same assay again) vs cold
for mode, shift in (("max warm", 0), ("max cold", 1)):
    ev = []
    for it in range(40):
        a = (it * shift) % A
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fmax(a); e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    t = np.array([a.elapsed_time(b) for a, b in ev][8:])
    print(f"{mode}: {np.median(t)*1e3:.1f} us median, {t.min()*1e3:.1f} min -> {(128 + 64) * 1.048576 / np.median(t):.0f} GB/s", flush=True)
