#!/usr/bin/env python3
"""Per-batch-size average durations of the cross-stack kernels (backward: two launches; forward) from a rocprofv3 kernel trace of
cross_bwd_probe.py (55 calls per batch size, the first 10 of each dropped)."""
import csv
import sys

allrows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_cross_" in r["Kernel_Name"]]
rows = [r for r in allrows if "k_cross_bwd" in r["Kernel_Name"]]
fwd = sorted((r for r in allrows if "k_cross_fwd" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sizes = [int(v) for v in sys.argv[2].split(",")]
main = [r for r in rows if "finish" not in r["Kernel_Name"]]
fin = [r for r in rows if "finish" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for i, B in enumerate(sizes):
    m = [dur(r) for r in main[55 * i + 10:55 * (i + 1)]]
    f = [dur(r) for r in fin[55 * i + 10:55 * (i + 1)]]
    fw = [dur(r) for r in fwd[55 * i + 10:55 * (i + 1)]] or [float("nan")]
    print(f"{sys.argv[3] if len(sys.argv) > 3 else ''} B = {B:6d}: k_cross_bwd {sum(m) / len(m):7.1f} us   k_cross_bwd_finish {sum(f) / len(f):6.1f} us   "
          f"k_cross_fwd4 {sum(fw) / len(fw):6.1f} us")
