set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases_gpu.py tests/test_finish_gpu.py tests/test_wide_deep_gpu.py tests/test_shard_gpu.py -x -q > gpurun_out/r05_t3.log 2>&1 || { tail -30 gpurun_out/r05_t3.log; exit 1; }
tail -2 gpurun_out/r05_t3.log
bash tools/probes/ab_libs.sh "$@"
