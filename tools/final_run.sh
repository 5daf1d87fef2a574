set -e
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_line.json 2> gpurun_out/final/bench.err
echo "bench done"; cut -c1-200 gpurun_out/final/bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/final/bench_line_under_rocprof.json 2> gpurun_out/final/prof.err
cp $(find gpurun_out/final/prof -name "*kernel_stats.csv") gpurun_out/final/kernel_stats.csv
python tools/prof_summary.py gpurun_out/final/kernel_stats.csv > gpurun_out/final/kernel_summary.txt
echo "prof done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/final/pmc_fetch -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/final/pmc1.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/final/pmc_write -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2> gpurun_out/final/pmc2.err
python tools/pmc_summary.py gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write gpurun_out/final/pmc_traffic.json > gpurun_out/final/pmc_traffic.txt
echo "pmc done"; cat gpurun_out/final/pmc_traffic.txt
python bench.py --no-cpu-baseline --dist zipf --fields 39 > gpurun_out/final/bench_line_zipf39.json 2>/dev/null
python tools/paths_bench.py > gpurun_out/final/paths_bench.txt 2>/dev/null
python tools/dcn_bench.py > gpurun_out/final/dcn_bench.txt 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/crossprof -- python3 tools/cross_probe.py > /dev/null 2>&1
python tools/prof_summary.py $(find gpurun_out/final/crossprof -name "*kernel_stats.csv") > gpurun_out/final/cross_kernel_summary.txt
cat gpurun_out/final/paths_bench.txt gpurun_out/final/dcn_bench.txt

# id-distribution sweep: uniform / Zipf(1.05), 26 / 39 fields
for d in uniform zipf; do for f in 26 39; do
  python bench.py --no-cpu-baseline --dist $d --fields $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$d', $f, 'fields:', d['value'], 'samples/s', d['ms_per_step'], 'ms/step, U/N', d['config']['unique_frac'], ', apply', d['roofline']['avg_ms'], 'ms', d['roofline']['achieved'], 'GB/s')"
done; done > gpurun_out/final/dist_sweep.txt
cat gpurun_out/final/dist_sweep.txt
rm -rf gpurun_out/final/prof gpurun_out/final/crossprof
