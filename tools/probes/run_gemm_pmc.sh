set -e
mkdir -p gpurun_out
timeout -k 10 250 ./tools/probes/dense_gemm_test > gpurun_out/gemm3.log 2>&1 || echo "probe rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_a -- $GRAFT_REPO_ROOT/tools/probes/dense_gemm_test 16384 prof > $GRAFT_REPO_ROOT/gpurun_out/pmc_a.log 2>&1 || echo "pmc a rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_b -- $GRAFT_REPO_ROOT/tools/probes/dense_gemm_test 16384 prof > $GRAFT_REPO_ROOT/gpurun_out/pmc_b.log 2>&1 || echo "pmc b rc=$?"
cd $GRAFT_REPO_ROOT
tail -45 gpurun_out/gemm3.log
find gpurun_out/pmc_a gpurun_out/pmc_b -name "*.csv" | head
