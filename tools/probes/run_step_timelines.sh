set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tl
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 10 --warmup 5 --repeats 1 --no-cpu-baseline --no-zipf39 --stamps never > $O/line.json 2> $O/err.log
for k in 0 1 2 3 4; do echo "== step +$k"; python3 $R/tools/step_timeline.py $O/trace $k; done > $O/timelines.txt
rm -rf $O/trace
