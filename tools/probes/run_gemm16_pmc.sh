# SQ counters of the 16-bit DenseLayer kernels INSIDE the benchmarked step (k_gemm256, k_gemm256_bwd, k_tail): two rocprofv3 --pmc
# passes over bench.py (counter collection serialises the kernels: each is measured running alone), per-kernel means.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gemm16_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 2 --repeats 1 --prime-steps 0 --no-cpu-baseline --no-zipf39"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/a -- $B > $O/a.log 2>&1 || echo "pmc a rc=$?"
timeout -k 10 280 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/b -- $B > $O/b.log 2>&1 || echo "pmc b rc=$?"
cd $R
{
echo "# rocprofv3 --pmc over bench.py (fp16 DenseLayers of the benchmarked step; counter collection runs every kernel alone), per-kernel means"
for f in "k_gemm256<" "k_gemm256_bwd<" "k_tail<"; do python3 tools/probes/pmc_table.py $O/a "$f"; python3 tools/probes/pmc_table.py $O/b "$f"; done
} > $O/gemm16_pmc.txt
rm -rf $O/a $O/b
cat $O/gemm16_pmc.txt
