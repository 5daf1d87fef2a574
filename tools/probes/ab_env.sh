# A/B of environment settings on ONE box: bash tools/probes/ab_env.sh "" "MREC_X=1" ...   (each argument: a space-separated VAR=value list, "" = none)
for e in "$@"; do
  env $e python bench.py --no-cpu-baseline 2>gpurun_out/ab_env_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); z=d.get('roofline_zipf39',{}); e=d.get('roofline_embedding_path',{})
print('[$e]', 'step', d['ms_per_step'], 'apply', d['roofline']['avg_ms'], 'lookup', e.get('lookup_ms'), 'apply_all', e.get('apply_ms_incl_finishing_kernel'), 'path', e.get('frac'), '| zipf39 step', z.get('ms_per_step'), 'apply', z.get('avg_ms'))" || tail -5 gpurun_out/ab_env_err.log
done
