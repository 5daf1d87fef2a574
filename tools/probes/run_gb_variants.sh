R=${GRAFT_REPO_ROOT:-$(pwd)}
for lib in libmrec_hip.so libmrec_gb2.so libmrec_gb8.so libmrec_hip.so; do
  export MREC_HIP_LIB=$R/mindrec_amd/csrc/$lib
  python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys;d=json.loads(sys.stdin.read());e=d['roofline_embedding_path'];z=d['roofline_zipf39'];print('$lib','step',d['ms_per_step'],'lookup',e['lookup_ms'],'emb',e['frac'],'z39 lookup',z['embedding_path']['lookup_ms'],'z39 step',z['ms_per_step'])"
done
