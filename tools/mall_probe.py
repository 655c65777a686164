"""Infinity Cache reuse between two streaming passes over one 134 MB block (mg_stream_probe, one 16-byte access per
lane): a read pass over block i directly followed by a copy pass reading block i (reuse) -- or block i + 8 (none).
Run under rocprofv3 --kernel-trace; the copy kernels' durations tell.  python tools/mall_probe2.py [--mb 128]"""
import argparse
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from magnify_amd import _native as nat, hotpath as hp


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=128)
    args = ap.parse_args()
    n = args.mb << 20
    N = 16
    src = torch.randint(0, 255, (N, n), dtype=torch.uint8, device="cuda")
    dst = torch.empty((N, n), dtype=torch.uint8, device="cuda")
    sink = torch.zeros(4, dtype=torch.int32, device="cuda")
    lib = nat.lib()

    def probe(s, d, mode, nb=n):
        nat.check(lib.mg_stream_probe(s.data_ptr(), d.data_ptr(), nb, mode, sink.data_ptr(), 0, hp._stream()), "probe")

    ev = lambda: torch.cuda.Event(enable_timing=True)
    for name, shift in (("reuse", 0), ("none", 8)):
        for rep in range(3):
            tot = 0.0
            for i in range(N):
                probe(src[(i + shift) % N], dst[0], 1)
                a, b = ev(), ev()
                a.record()
                probe(src[i], dst[i], 0)
                b.record()
                b.synchronize()
                tot += a.elapsed_time(b)
            print(name, "copy after read: %.1f us per block of %d MiB (events)" % (1e3 * tot / N, args.mb))
    # half-block granularity: read 2 halves, copy 2 halves
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
