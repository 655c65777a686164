"""One steady-state step of bench.py as the GPU ran it, from a rocprofv3 --kernel-trace CSV: every kernel in order with
its duration and the idle time before it, and the totals per kernel name.
usage: python tools/step_timeline.py <dir with *kernel_trace.csv> [step index from the end = 5] [first kernel of a step]
(default first kernel: the flat-field maxima pass; e.g. k_plane_minmax for a call of mg.beads without flat-field)"""
import collections
import csv
import glob
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([\w:]+(?:<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:64]


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rows = []
    for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    first = sys.argv[3] if len(sys.argv) > 3 else None
    starts = [i for i, r in enumerate(rows) if (first in r[2] if first else ("k_flatfield_max" in r[2] or "k_flat_rcmax" in r[2]))]
    starts = [i for k, i in enumerate(starts) if k == 0 or i - starts[k - 1] > 3]  # (rcmax + lean: one step start)
    if len(starts) < back + 2:
        print("too few steps in the trace")
        return
    a, b = starts[-back - 1], starts[-back]
    seg = rows[a:b]
    span = rows[b][0] - rows[a][0]
    busy = sum(e - s for s, e, _ in seg)
    print(f"step of {len(seg)} kernels: span {span / 1e3:.1f} us, busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us")
    end = seg[0][0]
    agg = collections.OrderedDict()
    for s, e, n in seg:
        k = short(n)
        print(f"  +{max(0, s - end) / 1e3:6.1f} idle  {(e - s) / 1e3:8.1f} us  {k}")
        end = max(end, e)
        v = agg.setdefault(k, [0, 0])
        v[0] += e - s
        v[1] += 1
    print("totals")
    for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print(f"  {t / 1e3:8.1f} us  x{c:<3d} {k}")


if __name__ == "__main__":
    main()
