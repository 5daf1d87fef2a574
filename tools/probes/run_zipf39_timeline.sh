set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/zipf39_prof
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --dist zipf --fields 39 --steps 30 --warmup 5 --no-cpu-baseline --no-zipf39 --stamps always > $O/bench_line_zipf39_under_rocprof.json 2> $O/prof.err
cp $(find $O/prof -name "*kernel_stats.csv") $O/kernel_stats.csv
python3 $R/tools/prof_summary.py $O/kernel_stats.csv > $O/kernel_summary.txt
python3 $R/tools/step_timeline.py $O/prof > $O/step_timeline.txt 2>&1
rm -rf $O/prof
head -30 $O/step_timeline.txt
