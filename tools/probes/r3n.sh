mkdir -p gpurun_out/r3n
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES --output-format csv -d $R/gpurun_out/r3n/pmc1 -- python3 $R/tools/probes/gemm32_probe.py 3 > $R/gpurun_out/r3n/pmc1.log 2>&1
python3 $R/tools/probes/pmc_table.py $R/gpurun_out/r3n/pmc1 gemm_f32 > $R/gpurun_out/r3n/pmc1.txt 2>&1; cat $R/gpurun_out/r3n/pmc1.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3n/pmc2 -- python3 $R/tools/probes/gemm32_probe.py 3 > $R/gpurun_out/r3n/pmc2.log 2>&1
python3 $R/tools/probes/pmc_table.py $R/gpurun_out/r3n/pmc2 gemm_f32 > $R/gpurun_out/r3n/pmc2.txt 2>&1; cat $R/gpurun_out/r3n/pmc2.txt
rm -rf $R/gpurun_out/r3n/pmc1 $R/gpurun_out/r3n/pmc2
