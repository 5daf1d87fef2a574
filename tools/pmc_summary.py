#!/usr/bin/env python3
"""Per-launch HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes: on gfx950 FETCH_SIZE tallies
128-B read requests at 64 B, so read bytes = 2 * FETCH_SIZE KiB; WRITE_SIZE is exact.

The correction is checked in the same runs on two launches whose traffic is known exactly:
  * the 1-GiB device-to-device copy bench.py times for `measured_copy_gbps` (reads 1 GiB, writes 1 GiB);
  * k_dense_adam4_slabs, from the byte counts bench.py prints in `dense_adam_bytes` (pass that JSON line as 4th argument).

usage: pmc_summary.py FETCH_DIR WRITE_DIR [OUT_JSON [BENCH_LINE_JSON]]"""
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


# key -> (substring of the demangled kernel name, required).  A required name that matches no launch is an ERROR (round 4 lost
# the lookup and dense-Adam rows -- and with them the calibration -- to two renamed kernels without anybody noticing).
NAMES = {"apply_main_adam": ("k_apply_main<4, int, UpdAdam", True),
         "gather_rows": ("k_gather_rows_w16<int", True),                      # the fused-row 16-bit lookup (deep row + wide word)
         "dense_adam": ("k_finish_dense_adam<", True),                         # dense Adam + the apply's finishing pass, one launch
         "dedup_insert": ("k_dedup_insert<int>", True),
         "gemm_fwd": ("k_gemm256<", True), "gemm_bwd": ("k_gemm256_bwd<", True), "tail": ("k_tail<", True),
         "copy_1gib": ("copyBuffer", True),
         # kernels of the non-default layouts (separate wide table, 4-byte stores): present only when those are benchmarked
         "gather_rows_w8": ("k_gather_rows<4, int", False), "apply_main_ftrl": ("k_apply_main<1, int, UpdFtrl", False),
         "wide_sum": ("k_wide_sum<int>", False), "head": ("k_head_fwd_bwd", False), "dense_adam_plain": ("k_dense_adam4", False)}


def main(fetch_dir, write_dir, out_json=None, bench_line=None):
    fe, wr = load(fetch_dir), load(write_dir)
    res = {}
    print(f"{'kernel':18s} {'launches':>8s} {'read MB (2*FETCH)':>18s} {'write MB':>10s} {'total MB':>10s}")
    missing = []
    for key, (pat, required) in NAMES.items():
        f = [x for k, v in fe.items() if pat in k for x in v]
        w = [x for k, v in wr.items() if pat in k for x in v]
        if not f or not w:
            if required:
                missing.append(f"{key} ({pat!r})")
            continue
        if key == "copy_1gib":                       # keep the large copies only (the 1-GiB ones of bench.py)
            f = [x for x in f if x > 0.25 * max(f)]
            w = [x for x in w if x > 0.25 * max(w)]
        elif key not in ("gemm_fwd", "gemm_bwd"):
            f, w = f[2:], w[2:]                      # skip the first 2 launches (warm-up steps)
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        rd, wt = 2 * fk * 1024, wk * 1024
        res[key] = {"read_bytes": rd, "write_bytes": wt, "total_bytes": rd + wt, "launches": len(f)}
        print(f"{key:18s} {len(f):8d} {rd / 1e6:18.1f} {wt / 1e6:10.1f} {(rd + wt) / 1e6:10.1f}")
    cal = {}
    if "copy_1gib" in res:
        c = res["copy_1gib"]
        cal["copy_1gib"] = {"read_ratio": c["read_bytes"] / 2 ** 30, "write_ratio": c["write_bytes"] / 2 ** 30}
    if bench_line and "dense_adam" in res:
        exp = json.loads(open(bench_line).read().strip().splitlines()[-1]).get("dense_adam_bytes")
        if exp:
            c = res["dense_adam"]
            cal["dense_adam"] = {"read_ratio": c["read_bytes"] / exp["read"], "write_ratio": c["write_bytes"] / exp["write"],
                                 "expected_read": exp["read"], "expected_write": exp["write"]}
    for k, c in cal.items():
        print(f"calibration on {k}: read {c['read_ratio']:.4f} x expected, write {c['write_ratio']:.4f} x expected")
    res["calibration"] = cal
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)
    if missing:
        names = sorted({k for k in fe})
        sys.exit("pmc_summary: no launch matched " + ", ".join(missing) + " -- a kernel was renamed; fix NAMES.  Kernels seen:\n  "
                 + "\n  ".join(names))
    if bench_line and "dense_adam" not in cal:
        sys.exit("pmc_summary: the dense-Adam calibration is missing (no dense_adam_bytes in the bench line)")


if __name__ == "__main__":
    main(*sys.argv[1:])
