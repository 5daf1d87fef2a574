# weight-gradient slab counts per layer (bench.py --dw-slabs layer:S,...) against the step time; "" = the library's proposal
for cfg in "" "0:7,1:16,2:32,3:64" "0:6,1:6,2:16,3:32" "0:6,1:8,2:16,3:16" "0:6,1:8,2:16,3:64"; do
python bench.py --no-cpu-baseline ${cfg:+--dw-slabs $cfg} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('slabs [$cfg]', d['ms_per_step'], d['ms_per_step_min'], d['ms_per_step_max'], d['kernels_ms']['mlp_fwd_bwd'], d['kernels_ms']['apply_dense'])"
done
