mkdir -p gpurun_out/r3n
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/probes/move_rows_check.py > $R/gpurun_out/r3n/mv.log 2>&1; tail -1 $R/gpurun_out/r3n/mv.log
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3n/kt -- python3 $R/tools/probes/gemm32_probe.py 10 > $R/gpurun_out/r3n/kt.log 2>&1
python3 $R/tools/prof_summary.py $(find $R/gpurun_out/r3n/kt -name "*kernel_stats.csv") | grep -i "gemm_f32" | cut -c1-200
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r3n/pmc1 -- python3 $R/tools/probes/gemm32_probe.py 3 > $R/gpurun_out/r3n/pmc1.log 2>&1
python3 $R/tools/probes/pmc_table.py $R/gpurun_out/r3n/pmc1 gemm_f32 > $R/gpurun_out/r3n/pmc1.txt 2>&1; cat $R/gpurun_out/r3n/pmc1.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3n/pmc2 -- python3 $R/tools/probes/gemm32_probe.py 3 > $R/gpurun_out/r3n/pmc2.log 2>&1
python3 $R/tools/probes/pmc_table.py $R/gpurun_out/r3n/pmc2 gemm_f32 > $R/gpurun_out/r3n/pmc2.txt 2>&1; cat $R/gpurun_out/r3n/pmc2.txt
rm -rf $R/gpurun_out/r3n/kt $R/gpurun_out/r3n/pmc1 $R/gpurun_out/r3n/pmc2
cd $R && python -m pytest tests/test_feature_cache_gpu.py -q > gpurun_out/r3n/t_cache.log 2>&1; tail -3 gpurun_out/r3n/t_cache.log
