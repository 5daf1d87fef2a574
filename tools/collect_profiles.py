#!/usr/bin/env python3
"""Copies the outputs of tools/final_run.sh (gpurun_out/final/) into profiles/ under a per-round prefix."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = ["bench_line.json", "bench_line_bf16.json", "deepfm_bench.txt", "cache_bench.txt", "bench_line_under_rocprof.json", "bench_kernel_stats.csv", "bench_kernel_summary.txt",
        "step_timeline_under_rocprof.txt", "pmc_traffic.json", "pmc_traffic.txt", "bench_line_zipf39.json", "paths_bench.txt",
        "dcn_bench.txt", "dense_gemm_probe.txt", "dist_sweep.txt", "bench_line_shard_protocol.json", "embed_folded.txt", "bench_line_dropout.json", "tail_probe.txt", "bench_line_fp32net.json",
        "x3_gemm_bench.txt", "published_config_line.json", "clock_offsets.json", "bench_line_shard_protocol_zipf39.json",
        "bench_line_shard_protocol_zipf39_uniques.json", "config5_line.json", "gemm16_pmc.txt", "map_pmc.txt"]


def main(prefix):
    src = os.path.join(ROOT, "gpurun_out", "final")
    for name in KEEP:
        p = os.path.join(src, name)
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(ROOT, "profiles", f"{prefix}_{name}"))
            print("kept", name)
        else:
            print("MISSING", name)


if __name__ == "__main__":
    main(sys.argv[1])
