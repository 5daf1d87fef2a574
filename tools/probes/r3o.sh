mkdir -p gpurun_out/r3o
for bm in 0 64 128; do echo "MREC_GF32_BM=$bm"; MREC_GF32_BM=$bm python tools/dcn_bench.py 2>/dev/null | tail -4; done > gpurun_out/r3o/dcn_bm.txt 2>&1; cat gpurun_out/r3o/dcn_bm.txt
python -m pytest tests/test_dense32_gpu.py tests/test_feature_cache_gpu.py -q -x > gpurun_out/r3o/t.log 2>&1; tail -3 gpurun_out/r3o/t.log
for v in 0 1; do MREC_ADAM_SIDE=$( [ $v = 1 ] && echo 1 ) python bench.py --no-cpu-baseline --no-zipf39 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('adam_side $v', d['ms_per_step'], d['ms_per_step_min'])"; done > gpurun_out/r3o/adam_side.txt 2>&1; cat gpurun_out/r3o/adam_side.txt
bash tools/aw_sweep.sh > gpurun_out/r3o/aw_sweep.txt 2>&1; cat gpurun_out/r3o/aw_sweep.txt
