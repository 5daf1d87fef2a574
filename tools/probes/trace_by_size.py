#!/usr/bin/env python3
"""Average duration per kernel name and per consecutive group of `reps` calls from a rocprofv3 kernel trace (argv: csv, reps,
comma-separated labels of the groups, substrings of the kernel names to keep)."""
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
reps, labels, keep = int(sys.argv[2]), sys.argv[3].split(","), sys.argv[4].split(",")
for name in keep:
    sel = [r for r in rows if name in r["Kernel_Name"]]
    for i, lab in enumerate(labels):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in sel[reps * i + 5:reps * (i + 1)]]
        if d:
            print(f"{name:24s} {lab:>8s}: {sum(d) / len(d):8.1f} us")
