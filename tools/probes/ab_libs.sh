# A/B of libmrec_hip builds on ONE box: bash tools/probes/ab_libs.sh libA.so libB.so ...   (files in mindrec_amd/csrc/)
for lib in "$@"; do
  MREC_HIP_LIB=$PWD/mindrec_amd/csrc/$lib python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); z=d.get('roofline_zipf39',{}); e=d.get('roofline_embedding_path',{})
print('$lib', 'step', d['ms_per_step'], 'apply', d['roofline']['avg_ms'], 'lookup', e.get('lookup_ms'), 'apply_all', e.get('apply_ms_incl_finishing_kernel'), 'path', e.get('frac'), '| zipf39 step', z.get('ms_per_step'), 'apply', z.get('avg_ms'), 'emb', z.get('embedding_path'))"
done
