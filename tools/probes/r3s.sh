mkdir -p gpurun_out/r3s
python -m pytest tests/test_dense32_gpu.py -q -x > gpurun_out/r3s/t.log 2>&1; tail -3 gpurun_out/r3s/t.log
python tools/dcn_bench.py 2>/dev/null | tail -4 > gpurun_out/r3s/dcn.txt; cat gpurun_out/r3s/dcn.txt
python -m pytest tests/test_wide_deep_gpu.py tests/test_bench_shape_gpu.py -q -k "dcn or deep_cross or cross" > gpurun_out/r3s/t_dcn.log 2>&1; tail -2 gpurun_out/r3s/t_dcn.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
