#!/bin/bash
# apply-kernel time vs table size (same N, same U/N): separates address-translation / DRAM-locality effects from bandwidth
for v in 200000000 50000000 10000000 2000000; do
  python bench.py --no-cpu-baseline --vocab $v > gpurun_out/ab.log 2>&1
  tail -1 gpurun_out/ab.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('vocab', $v, 'ms/step', d['ms_per_step'], 'apply ms', d['roofline']['avg_ms'], 'frac', d['roofline']['frac'], 'U/N', d['config']['unique_frac'], 'gather ms', d['kernels_ms']['gather_deep'])"
done
