"""GPU idle gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV.
usage: python tools/gap_report.py <dir with *kernel_trace.csv> [n_last_steps=3]
Prints the total idle time of the tail of the trace and the largest gaps with the kernels around them."""
import csv
import glob
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed region: everything after the last long pause (> 50 ms: CPU baseline / setup)
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 50_000_000:
        cut = i
rows = rows[cut:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
print(f"kernels {len(rows)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms")
gaps = []
end = rows[0][1]
for i in range(1, len(rows)):
    g = rows[i][0] - end
    if g > 0:
        gaps.append((g, rows[i - 1][2][:60], rows[i][2][:60]))
    end = max(end, rows[i][1])
agg = {}
for g, a, b in gaps:
    k = (a, b)
    agg[k] = agg.get(k, [0, 0])
    agg[k][0] += g
    agg[k][1] += 1
for (a, b), (g, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{g / 1e6:8.3f} ms  x{n:<4d} {a}  ->  {b}")
