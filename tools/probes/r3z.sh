mkdir -p gpurun_out/r3z
python -m pytest tests/test_dense_gpu.py tests/test_dense32_gpu.py tests/test_tail_gpu.py -q -x > gpurun_out/r3z/t.log 2>&1; tail -2 gpurun_out/r3z/t.log
python tools/dcn_bench.py 2>/dev/null | tail -4 > gpurun_out/r3z/dcn.txt; cat gpurun_out/r3z/dcn.txt
for v in 0 1; do
  if [ $v = 1 ]; then export MREC_FWD_MR4=1; fi
  python bench.py --no-cpu-baseline --no-zipf39 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwd_mr4=$v', d['ms_per_step'], d['ms_per_step_min'], d['kernels_ms']['mlp_fwd_bwd'])"
done > gpurun_out/r3z/ab.txt 2>&1; cat gpurun_out/r3z/ab.txt
unset MREC_FWD_MR4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3z/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-zipf39 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3z/prof.err
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $GRAFT_REPO_ROOT/gpurun_out/r3z/prof > $GRAFT_REPO_ROOT/gpurun_out/r3z/timeline.txt 2>&1; cut -c1-110 $GRAFT_REPO_ROOT/gpurun_out/r3z/timeline.txt | head -30
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3z/prof
