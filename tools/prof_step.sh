# one profiled bench run: per-kernel stats + the timeline of one step (run on the GPU box from the repo root)
set -e
OUT=${1:-gpurun_out/p1}
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -- python3 $R/bench.py --steps 20 --warmup 3 --repeats 2 --no-cpu-baseline $BENCH_FLAGS > $R/$OUT/bench_line_under_rocprof.json 2> $R/$OUT/prof.err
cd $R
cp $(find $OUT/prof -name "*kernel_stats.csv") $OUT/kernel_stats.csv
python tools/prof_summary.py $OUT/kernel_stats.csv > $OUT/kernel_summary.txt
python tools/step_timeline.py $OUT/prof > $OUT/step_timeline.txt 2>&1 || true
rm -rf $OUT/prof
