#!/bin/bash
# apply-kernel window / batch-depth sweep: libraries built with EXTRA_FLAGS="-DMREC_AW4=.. -DMREC_AB4=.. -DMREC_GP4=.." OUT=libmrec_awX_abY_gpZ.so
for args in "" "--dist zipf --fields 39"; do
for lib in "" $(ls mindrec_amd/csrc/libmrec_aw*.so); do
  MREC_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 20 $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$args]', '${lib:-default}', 'apply', d['roofline']['avg_ms'], 'ms  step', d['ms_per_step'])"
done; done
