# Hot columns off / constant columns only / constant + dominant ids, per id distribution (the engine reads MREC_CONST_COLS and
# MREC_HOT_MIN), ONE box: bash tools/probes/ab_const_cols.sh
# (the arms run the SAME library: compare a line of the 26-field case with the committed round's profile too -- a change that slows the
# kernel in every arm does not show here)
for cfg in "--dist uniform --fields 26" "--dist zipf --fields 39" "--dist uniform --fields 39" "--dist zipf --fields 26"; do
for arm in "0 0" "1 0" "1 1024" "0 0" "1 0" "1 1024"; do
set -- $arm
MREC_CONST_COLS=$1 MREC_HOT_MIN=$2 python bench.py --no-cpu-baseline --no-zipf39 --repeats 3 --stamps always $cfg 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d.get('roofline_embedding_path',{})
print('hot=$1 min=$2 $cfg :', 'step', d['ms_per_step'], 'apply_main', d['roofline']['avg_ms_stamps'], 'lookup', e.get('lookup_ms'), 'apply_all', e.get('apply_ms_incl_finishing_kernel'), 'finishing', e.get('finishing_pass_ms'), 'path', e.get('frac'), 'U/N', d['config']['unique_frac'])"
done; done
