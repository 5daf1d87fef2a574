set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp
python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
python3 $R/bench.py --mlp-dtype bf16 --no-cpu-baseline --no-zipf39 > $O/bench_line_bf16.json 2>> $O/bench.err
python3 $R/bench.py --dropout --no-cpu-baseline --no-zipf39 > $O/bench_line_dropout.json 2>> $O/bench.err
cd $R
python bench.py --no-cpu-baseline --shard-protocol 2>/dev/null | grep "^{\"metric\"" | tail -1 > $O/bench_line_shard_protocol.json
for f in bench_line bench_line_bf16 bench_line_dropout bench_line_shard_protocol; do python3 -c "
import json;d=json.load(open('$O/$f.json'));print('$f',d['value'],d['ms_per_step'],d.get('ms_per_step_min'),d.get('ms_per_step_max'))"; done
bash tools/final_run.sh 3 > $O/part3.log 2>&1; tail -16 $O/part3.log | cut -c1-200
