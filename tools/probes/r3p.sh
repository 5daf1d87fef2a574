mkdir -p gpurun_out/r3p
python bench.py --no-cpu-baseline --no-zipf39 > gpurun_out/r3p/bench.out 2> gpurun_out/r3p/bench.err; tail -5 gpurun_out/r3p/bench.err; cut -c1-150 gpurun_out/r3p/bench.out
MREC_ADAM_SIDE=1 python bench.py --no-cpu-baseline --no-zipf39 > gpurun_out/r3p/bench1.out 2> gpurun_out/r3p/bench1.err; tail -5 gpurun_out/r3p/bench1.err; cut -c1-150 gpurun_out/r3p/bench1.out
python -m pytest tests/test_feature_cache_gpu.py -q -x > gpurun_out/r3p/t.log 2>&1; tail -3 gpurun_out/r3p/t.log
