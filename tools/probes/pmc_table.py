#!/usr/bin/env python3
"""Per-kernel means of the counters in a rocprofv3 --pmc output directory: python pmc_table.py DIR [name-filter]"""
import csv
import glob
import sys
from collections import defaultdict

d, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if flt in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} n={len(v):3d}  mean {sum(v) / len(v):16.1f}")
