#!/usr/bin/env python3
"""Per-launch HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: on gfx950 FETCH_SIZE tallies
128-B read requests at 64 B, so read bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE is exact.  The
correction is checked in the same runs on k_dense_adam4, whose traffic is known exactly
(4 fp32 reads; 3 fp32 writes + the 2-byte bf16 shadow of the 2,820,097-float dense parameter buffer)."""
import collections
import csv
import glob
import json
import sys


def load(d):
    f = glob.glob(f"{d}/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")].append(float(r["Counter_Value"]))
    return agg


def main(fetch_dir, write_dir, out_json=None):
    fe, wr = load(fetch_dir), load(write_dir)
    names = {"apply_main_adam": "k_apply_main<4, int, UpdAdam", "gather_rows": "k_gather_rows<4, int",
             "apply_main_ftrl": "k_apply_main<1, int, UpdFtrl", "wide_sum": "k_wide_sum<int>",
             "dense_adam4": "k_dense_adam4", "dedup_insert": "k_dedup_insert<int>"}
    res = {}
    print(f"{'kernel':18s} {'launches':>8s} {'read MB (2*FETCH)':>18s} {'write MB':>10s} {'total MB':>10s}")
    for key, pat in names.items():
        f = [v for k, v in fe.items() if pat in k]
        w = [v for k, v in wr.items() if pat in k]
        if not f or not w:
            continue
        fk = sum(f[0][2:]) / len(f[0][2:])          # skip the first 2 launches (warm-up steps)
        wk = sum(w[0][2:]) / len(w[0][2:])
        rd, wt = 2 * fk * 1024, wk * 1024
        res[key] = {"read_bytes": rd, "write_bytes": wt, "total_bytes": rd + wt, "launches": len(f[0]) - 2}
        print(f"{key:18s} {len(f[0]) - 2:8d} {rd / 1e6:18.1f} {wt / 1e6:10.1f} {(rd + wt) / 1e6:10.1f}")
    n = 2820097 * 4
    c = res["dense_adam4"]
    # k_dense_adam4<true> also writes the bf16 operand shadow: 3 fp32 + 1 bf16 word per element
    print(f"calibration on k_dense_adam4<true>: read {c['read_bytes'] / (4 * n):.4f} x expected, write {c['write_bytes'] / (3.5 * n):.4f} x expected")
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:])
