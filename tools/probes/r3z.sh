mkdir -p gpurun_out/r3z
python -m pytest tests/test_dense_gpu.py tests/test_tail_gpu.py tests/test_bench_shape_gpu.py -q -x --deselect tests/test_bench_shape_gpu.py::test_auc_parity_on_the_benchmarked_path > gpurun_out/r3z/t.log 2>&1; tail -2 gpurun_out/r3z/t.log
python bench.py --no-cpu-baseline --no-zipf39 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['ms_per_step_min'], d['kernels_ms']['mlp_fwd_bwd'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3z/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-zipf39 > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3z/prof.err
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py $GRAFT_REPO_ROOT/gpurun_out/r3z/prof > $GRAFT_REPO_ROOT/gpurun_out/r3z/timeline.txt 2>&1; grep "gemm\|tail" $GRAFT_REPO_ROOT/gpurun_out/r3z/timeline.txt | cut -c1-110
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3z/prof
