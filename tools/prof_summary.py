#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace --stats kernel_stats.csv: short names, calls, avg us, total ms."""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    if n.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", n)
        return "hipblaslt_gemm " + n[:14] + " MT" + (m.group(1) if m else "?")
    n = re.sub(r"at::native::", "", n)
    return n[:110]


def main(path, steps=None):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print(f"{'kernel':112s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
    for r in sorted(rows, key=lambda r: -int(r["TotalDurationNs"])):
        t = int(r["TotalDurationNs"])
        print(f"{short(r['Name']):112s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.2f} {t / 1e6:10.3f} {100 * t / tot:6.2f}")
    print(f"total kernel time {tot / 1e6:.3f} ms")


if __name__ == "__main__":
    main(*sys.argv[1:])
