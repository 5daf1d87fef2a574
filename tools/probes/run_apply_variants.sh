#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r5n
for lib in libmrec_hip.so libmrec_aw4.so libmrec_aw16.so libmrec_hip.so; do
  export MREC_HIP_LIB=$R/mindrec_amd/csrc/$lib
  python3 $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $R/gpurun_out/r5n/$lib.json || exit 1
  python3 -c "
import json,sys;d=json.load(open('$R/gpurun_out/r5n/$lib.json'));print('$lib','step',d['ms_per_step'],'apply',d['roofline']['avg_ms'],'emb',d['roofline_embedding_path']['frac'],'zipf39 apply',d['roofline_zipf39']['avg_ms'],d['roofline_zipf39']['frac'])"
done
