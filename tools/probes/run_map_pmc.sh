# MapParameter.get over all-new keys (configs[4] shape): per-kernel durations and the L2 / atomic / wait counters of the insert chain
# (k_map_probe -> k_map_place -> k_map_finish) and the gather -- is k_map_place bound by its claims (device-scope atomics)?
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/map_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/map_probe.py new"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- $P > $O/t.log 2>&1 || echo "trace rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/a -- $P > $O/a.log 2>&1 || echo "pmc a rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum --output-format csv -d $O/c -- $P > $O/c.log 2>&1 || echo "pmc c rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $O/b -- $P > $O/b.log 2>&1 || echo "pmc b rc=$?"
cd $R
{
echo "# MapParameter.get, 425 984 all-new int64 keys, D = 128 (tools/map_probe.py new): rocprofv3 kernel statistics, then per-kernel counter means"
python3 tools/prof_summary.py $(find $O/t -name "*kernel_stats.csv") | head -12
for f in k_map_probe k_map_place k_map_finish k_gather_rows; do python3 tools/probes/pmc_table.py $O/a "$f"; python3 tools/probes/pmc_table.py $O/c "$f"; python3 tools/probes/pmc_table.py $O/b "$f"; done
} > $O/map_pmc.txt 2>&1
rm -rf $O/t $O/a $O/b $O/c
cat $O/map_pmc.txt
