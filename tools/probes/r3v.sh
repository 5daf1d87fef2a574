mkdir -p gpurun_out/r3v
python -m pytest tests/test_criteo_gpu.py -q -s > gpurun_out/r3v/criteo.log 2>&1; grep -E "max .diff|passed|failed|AUC" gpurun_out/r3v/criteo.log
for lib in "" mindrec_amd/csrc/libmrec_pre_w5.so mindrec_amd/csrc/libmrec_pre_w4.so mindrec_amd/csrc/libmrec_pre_w3.so; do
 for args in "" "--dist zipf --fields 39" "--dist zipf"; do
  MREC_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-zipf39 --steps 20 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$args]', '${lib:-default}', 'apply', d['roofline']['avg_ms'], 'ms  step', d['ms_per_step'], 'emb frac', d['roofline_embedding_path']['frac'])"
 done
done > gpurun_out/r3v/pre.txt 2>&1; cat gpurun_out/r3v/pre.txt
MREC_HIP_LIB=mindrec_amd/csrc/libmrec_pre_w4.so python -m pytest tests/test_gpu_parity.py tests/test_edge_cases_gpu.py tests/test_full_size_gpu.py -q -x > gpurun_out/r3v/parity_pre.log 2>&1; tail -2 gpurun_out/r3v/parity_pre.log
