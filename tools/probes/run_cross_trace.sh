#!/bin/bash
# Kernel-trace durations of the cross-stack backward's two launches over a range of batch sizes (one block on an idle chip
# up to the benchmark batch): tools/probes/cross_bwd_probe.py under rocprofv3, table by cross_trace_table.py.
set -e
OUT=${1:-gpurun_out/cross_trace}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/$OUT
export CROSS_PROBE_B=8,32,64,2048,16384,32768
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/xp
rocprofv3 --kernel-trace --output-format csv -d /tmp/xp -- python3 $R/tools/probes/cross_bwd_probe.py /tmp > $R/$OUT/probe.txt 2>&1
f=$(find /tmp/xp -name "*kernel_trace.csv" | head -1)
python3 $R/tools/probes/cross_trace_table.py "$f" $CROSS_PROBE_B > $R/$OUT/table.txt
cat $R/$OUT/table.txt
