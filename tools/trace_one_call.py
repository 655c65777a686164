"""Print the kernel timeline of the last call in a rocprofv3 --kernel-trace CSV (calls are delimited by the
first kernel whose name contains --first).  python tools/trace_one_call.py <kernel_trace.csv> [--first flatfield_max]"""
import argparse
import csv

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--first", default="flatfield_max")
ap.add_argument("--min-us", type=float, default=0.0)
args = ap.parse_args()
rows = sorted(csv.DictReader(open(args.csv)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if args.first in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev, tot = t0, 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    if (e - s) / 1e3 >= args.min_us or (s - prev) / 1e3 > 10:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev) / 1e3:7.1f}  {r['Kernel_Name'][:90]}")
    prev = e
print(f"kernels {tot / 1e3:.1f} us in {b - a} launches; call span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
