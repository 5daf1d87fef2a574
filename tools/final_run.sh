# Produces every measurement the profiles/ directory holds for a round, in three parts that each fit one gpurun call
# (run on the GPU box from the repo root):
#   bash tools/final_run.sh 1   bench lines (fp16 default with cpu_baseline, bf16, dropout), rocprof kernel stats + timeline, PMC traffic
#   bash tools/final_run.sh 2   shard protocol lines, id-distribution sweep, the embedding kernels alone
#   bash tools/final_run.sh 3   secondary paths: cross / FM / MapParameter, Deep&Cross and DeepFM steps, GEMM probes, cache tier
# then  python tools/collect_profiles.py rNN
set -e
PART=${1:-1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
if [ "$PART" = 1 ]; then
cd /tmp && export TMPDIR=/tmp
RND=${RND:-r05}
# the profiler's clock first: kernel statistics of the benchmarked command, and what rocprofv3 adds to the kernels' own stamps
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-zipf39 --stamps always > $O/bench_line_under_rocprof.json 2> $O/prof.err
cp $(find $O/prof -name "*kernel_stats.csv") $O/bench_kernel_stats.csv
python3 $R/tools/prof_summary.py $O/bench_kernel_stats.csv > $O/bench_kernel_summary.txt
python3 $R/tools/step_timeline.py $O/prof > $O/step_timeline_under_rocprof.txt 2>&1
python3 $R/tools/clock_offsets.py $O/bench_line_under_rocprof.json $O/bench_kernel_summary.txt $O/clock_offsets.json
cp $O/clock_offsets.json $R/profiles/${RND}_clock_offsets.json            # (the lines below are printed with THIS run's offsets and summary)
cp $O/bench_kernel_summary.txt $R/profiles/${RND}_bench_kernel_summary.txt
echo "prof done"
python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench done"; cut -c1-200 $O/bench_line.json
python3 $R/bench.py --mlp-dtype bf16 --no-cpu-baseline --no-zipf39 > $O/bench_line_bf16.json 2>> $O/bench.err
python3 $R/bench.py --dropout --no-cpu-baseline --no-zipf39 > $O/bench_line_dropout.json 2>> $O/bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --prime-steps 0 --no-cpu-baseline --no-zipf39 > $O/pmc_bench_line.json 2> $O/pmc1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 6 --warmup 2 --repeats 1 --prime-steps 0 --no-cpu-baseline --no-zipf39 > /dev/null 2> $O/pmc2.err
python3 $R/tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json $O/pmc_bench_line.json > $O/pmc_traffic.txt
echo "pmc done"; cat $O/pmc_traffic.txt
rm -rf $O/prof $O/pmc_fetch $O/pmc_write
fi
if [ "$PART" = 2 ]; then
cd $R
# what a rank of an N-GPU job does besides moving bytes over xGMI: the row-shard protocol over RCCL with itself, and its kernels alone
python bench.py --no-cpu-baseline --shard-protocol 2>/dev/null | grep "^{\"metric\"" | tail -1 > $O/bench_line_shard_protocol.json
cut -c1-220 $O/bench_line_shard_protocol.json
# ... on Criteo-like ids x 39 fields: one 16-bit row per position against one fp32 row per UNIQUE id on the wire (the line's rccl block carries
# the byte model of an 8-rank node for both)
python bench.py --no-cpu-baseline --shard-protocol --dist zipf --fields 39 2>/dev/null | grep "^{\"metric\"" | tail -1 > $O/bench_line_shard_protocol_zipf39.json
python bench.py --no-cpu-baseline --shard-protocol --dist zipf --fields 39 --shard-uniques 0.3 2>/dev/null | grep "^{\"metric\"" | tail -1 > $O/bench_line_shard_protocol_zipf39_uniques.json
python bench.py --no-cpu-baseline --dist zipf --fields 39 > $O/bench_line_zipf39.json 2>/dev/null
# id-distribution sweep: uniform / Zipf(1.05), 26 / 39 fields
for d in uniform zipf; do for f in 26 39; do
  python bench.py --no-cpu-baseline --no-zipf39 --repeats 3 --dist $d --fields $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', $f, 'fields:', d['value'], 'samples/s', d['ms_per_step'], 'ms/step, U/N', d['config']['unique_frac'], ', apply', d['roofline']['avg_ms'], 'ms', d['roofline']['achieved'], 'GB/s, embedding path', d.get('roofline_embedding_path', {}).get('frac'))"
done; done > $O/dist_sweep.txt
cat $O/dist_sweep.txt
python tools/embed_bench.py --layout folded --tag final 2>/dev/null > $O/embed_folded.txt
python tools/embed_bench.py --layout folded --dist zipf --fields 39 --tag zipf39 2>/dev/null >> $O/embed_folded.txt
cat $O/embed_folded.txt
fi
if [ "$PART" = 3 ]; then
cd $R
python tools/paths_bench.py > $O/paths_bench.txt 2>/dev/null
python tools/dcn_bench.py 50 x3 2>/dev/null | grep "DCN step" > $O/dcn_bench.txt
python tools/dcn_bench.py 50 exact 2>/dev/null >> $O/dcn_bench.txt
python tools/probes/x3_bench.py 2>/dev/null | grep "^M" > $O/x3_gemm_bench.txt
python tools/published_config_bench.py 2>/dev/null | tail -1 > $O/published_config_line.json
for dt in fp16 bf16 fp32; do python tools/deepfm_bench.py $dt 2>/dev/null | tail -1; done > $O/deepfm_bench.txt      # (a process per dtype: a second engine in one process runs slower)
cat $O/paths_bench.txt $O/dcn_bench.txt $O/deepfm_bench.txt
if [ -x ./tools/probes/dense_gemm_test ]; then timeout -k 10 200 ./tools/probes/dense_gemm_test > $O/dense_gemm_probe.txt 2>&1; fi
timeout -k 10 200 python tools/probes/tail_probe.py 2>/dev/null > $O/tail_probe.txt
timeout -k 10 300 python tools/cache_bench.py > $O/cache_bench.txt 2>/dev/null
tail -5 $O/cache_bench.txt
python bench.py --no-cpu-baseline --no-zipf39 --mlp-dtype fp32 2>/dev/null | tail -1 > $O/bench_line_fp32net.json      # the fp32-net option (exact-fp32 MFMA DenseLayers)
bash tools/probes/run_cross_trace.sh gpurun_out/final/cross_trace
python tools/config5_bench.py fp16 2>/dev/null | tail -1 > $O/config5_line.json
fi
if [ "$PART" = 4 ]; then
cd $R
# SQ counters of the 16-bit DenseLayer kernels inside the benchmarked step; L2 / atomic / wait counters of MapParameter.get over new keys
bash tools/probes/run_gemm16_pmc.sh > /dev/null 2>&1; cp gpurun_out/gemm16_pmc/gemm16_pmc.txt $O/gemm16_pmc.txt
bash tools/probes/run_map_pmc.sh > /dev/null 2>&1; cp gpurun_out/map_pmc/map_pmc.txt $O/map_pmc.txt
tail -30 $O/map_pmc.txt
fi
