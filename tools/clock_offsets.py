#!/usr/bin/env python3
"""rocprofv3's clock against the kernels' own stamps, where both saw the SAME launches: `bench.py` run under `rocprofv3 --kernel-trace
--stats` prints its stamp averages (roofline.avg_ms_stamps, roofline_lookup.avg_ms_stamps), rocprofv3's kernel statistics hold its
average dispatch durations.  The difference per kernel -- the time in front of the first workgroup and behind the last wave of a
dispatch -- is written to profiles/rNN_clock_offsets.json, which bench.py adds to its stamps so that `roofline.frac` reads by the
profiler's clock.

usage: clock_offsets.py BENCH_LINE_UNDER_ROCPROF.json KERNEL_SUMMARY.txt OUT.json"""
import json
import sys


def main(line_path, summary_path, out_path):
    d = json.loads(open(line_path).read().strip().splitlines()[-1])
    st_apply = d["roofline"].get("avg_ms_stamps") or d["roofline"]["avg_ms"]
    st_lookup = d["roofline_lookup"].get("avg_ms_stamps") or d["roofline_lookup"]["avg_ms"]
    avg = {}
    for ln in open(summary_path):
        for key in ("k_apply_main<", "k_gather_rows_w16<"):
            if ln.startswith(key) and key not in avg:
                avg[key] = float(ln.split()[-3])
    if len(avg) != 2:
        sys.exit(f"clock_offsets: kernels not found in {summary_path}: {sorted(avg)}")
    out = {"apply_main_us": round(avg["k_apply_main<"] - st_apply * 1e3, 3), "lookup_us": round(avg["k_gather_rows_w16<"] - st_lookup * 1e3, 3),
           "rocprof_avg_us": {"k_apply_main": avg["k_apply_main<"], "k_gather_rows_w16": avg["k_gather_rows_w16<"]},
           "stamps_avg_us": {"k_apply_main": round(st_apply * 1e3, 3), "k_gather_rows_w16": round(st_lookup * 1e3, 3)},
           "source": "one run of `rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-zipf39`: "
                     "its kernel statistics and the stamp averages the same process printed"}
    if not (0.0 <= out["apply_main_us"] <= 15.0 and 0.0 <= out["lookup_us"] <= 15.0):
        sys.exit(f"clock_offsets: implausible offsets {out}")
    json.dump(out, open(out_path, "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main(*sys.argv[1:4])
