"""Sum rocprofv3 --pmc counter CSVs per kernel: python tools/pmc_sum.py <dir> [kernel-substring]"""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in sorted(acc.items()):
    if flt in k:
        print(k, {a: f"{b:.4g}" for a, b in sorted(v.items())})
