set -e
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_edge_cases_gpu.py tests/test_finish_gpu.py tests/test_wide_deep_gpu.py tests/test_shard_gpu.py -x -q > gpurun_out/r05_t3.log 2>&1 || { tail -30 gpurun_out/r05_t3.log; exit 1; }
tail -3 gpurun_out/r05_t3.log
for lib in libmrec_hip.so; do
  MREC_HIP_LIB=$PWD/mindrec_amd/csrc/$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); z=d.get('roofline_zipf39',{}); e=d.get('roofline_embedding_path',{})
print('$lib', 'step', d['ms_per_step'], 'apply', d['roofline']['avg_ms'], 'lookup', e.get('lookup_ms'), 'apply_all', e.get('apply_ms_incl_finishing_kernel'), 'path', e.get('frac'), '| zipf39 step', z.get('ms_per_step'), 'apply', z.get('avg_ms'), 'emb', z.get('embedding_path'))"
done
