# HBM bytes per launch of the embedding kernels on the Criteo-like workload (Zipf ids x 39 fields): separate --pmc FETCH_SIZE / WRITE_SIZE passes
# of the bench command, corrected and summarised by tools/pmc_summary.py (gpurun_out/zipf39_pmc/pmc_traffic.txt)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/zipf39_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--dist zipf --fields 39 --steps 6 --warmup 2 --repeats 1 --prime-steps 0 --no-cpu-baseline --no-zipf39 --stamps always"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $ARGS > $O/pmc_bench_line.json 2> $O/pmc1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $ARGS > /dev/null 2> $O/pmc2.err
python3 $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc_bench_line.json > $O/pmc_traffic.txt
rm -rf $O/pmc_fetch $O/pmc_write
cat $O/pmc_traffic.txt
