"""Per-kernel means of the counters of one rocprofv3 --pmc pass (counter_collection.csv) next to the kernel
durations of the same run (kernel_trace.csv).  usage: python tools/pmc_table.py <dir> [kernel substring ...]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pats = sys.argv[2:]
cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]
    if pats and not any(p in k for p in pats):
        continue
    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in vals.items():
    n = max(len(v) for v in cs.values())
    line = f"{k[:48]:48s} n={n:4d}"
    if k in dur:
        line += f" us={sum(dur[k]) / len(dur[k]):9.1f}"
    print(line)
    for c, v in sorted(cs.items()):
        print(f"    {c:24s} {sum(v) / len(v):16.0f}")
