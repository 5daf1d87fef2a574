# one-off GPU session: new kernels' checks, DCN bench, apply-variant sweep, RCCL channel count
mkdir -p gpurun_out/r3m
python tools/probes/move_rows_check.py > gpurun_out/r3m/mv.log 2>&1; tail -1 gpurun_out/r3m/mv.log
python -m pytest tests/test_dense32_gpu.py tests/test_feature_cache_gpu.py -q > gpurun_out/r3m/t_dense32.log 2>&1; tail -2 gpurun_out/r3m/t_dense32.log
python -m pytest tests/test_wide_deep_gpu.py tests/test_bench_shape_gpu.py -q -k "dcn or deep_cross or cross or cache or host" > gpurun_out/r3m/t_dcn.log 2>&1; tail -2 gpurun_out/r3m/t_dcn.log
python tools/dcn_bench.py > gpurun_out/r3m/dcn_bench.txt 2>/dev/null; cat gpurun_out/r3m/dcn_bench.txt
python tools/cache_bench.py > gpurun_out/r3m/cache_bench.txt 2>&1; tail -3 gpurun_out/r3m/cache_bench.txt
bash tools/aw_sweep.sh > gpurun_out/r3m/aw_sweep.txt 2>&1; cat gpurun_out/r3m/aw_sweep.txt
for ch in 4 16 32; do
  NCCL_MIN_NCHANNELS=$ch NCCL_MAX_NCHANNELS=$ch python bench.py --no-cpu-baseline --no-zipf39 --shard-protocol 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('channels $ch', d['ms_per_step'], {k:v['median'] for k,v in d['kernels_ms'].items() if k.startswith('a2a') or k.startswith('all')})"
done > gpurun_out/r3m/nccl_channels.txt 2>&1; cat gpurun_out/r3m/nccl_channels.txt
