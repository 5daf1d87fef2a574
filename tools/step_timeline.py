#!/usr/bin/env python3
"""One steady-state step from a rocprofv3 kernel trace: kernels in start order with their queue, start offset,
duration and the idle gap since the previous kernel ended anywhere on the device.  argv[2] (optional): how many steps behind the
middle one to print (a sink of 5 steps replays as one graph: the step behind a graph launch starts with the staging copy)."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if "k_apply_main<4" in r["Kernel_Name"] and "UpdAdam" in r["Kernel_Name"]]
k = len(marks) // 2 + (int(sys.argv[2]) if len(sys.argv) > 2 else 0)      # one step from the middle of the run (the timed region)
a, b = marks[k], marks[k + 1]
t0 = rows[a]["s"]
last_end = rows[a]["s"]
busy = 0
print(f"step length {(rows[b]['s'] - t0) / 1e3:.1f} us")
for r in rows[a:b]:
    gap = (r["s"] - last_end) / 1e3
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print(f"{(r['s'] - t0) / 1e3:8.1f}  dur {(r['e'] - r['s']) / 1e3:7.1f}  gap {gap:6.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    last_end = max(last_end, r["e"])
