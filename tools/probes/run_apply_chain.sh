#!/bin/bash
# One wave's chain in the sparse apply / the lookup: kernel-trace durations at batches of 4 .. 16384 samples (a handful of
# workgroups on an idle chip up to the benchmark batch), uniform ids x 26 fields and Zipf(1.05) x 39 fields.
set -e
OUT=${1:-gpurun_out/apply_chain}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
cd /tmp && export TMPDIR=/tmp
for c in "uniform 26" "zipf 39"; do
  set -- $c
  rm -rf /tmp/ap
  rocprofv3 --kernel-trace --output-format csv -d /tmp/ap -- python3 $R/tools/probes/apply_small_probe.py $1 $2 4,64,1024,16384 > /dev/null 2>&1
  echo "== $1 ids x $2 fields"
  python3 $R/tools/probes/trace_by_size.py $(find /tmp/ap -name "*kernel_trace.csv") 30 4,64,1024,16384 k_apply_main,k_apply_long,k_gather_rows
done > $R/$OUT/table.txt 2>&1
cat $R/$OUT/table.txt
