mkdir -p gpurun_out/r3w
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && python -m pytest tests/test_criteo_gpu.py -q -s > gpurun_out/r3w/criteo.log 2>&1; grep -E "diff|passed|failed|AUC" gpurun_out/r3w/criteo.log)
export MREC_HIP_LIB=$R/mindrec_amd/csrc/libmrec_pre_w4.so
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/r3w/pmc1 -- python3 $R/tools/embed_bench.py --layout folded --dist zipf --fields 39 --tag z > $R/gpurun_out/r3w/pmc1.log 2>&1
python3 $R/tools/probes/pmc_table.py $R/gpurun_out/r3w/pmc1 k_apply_main > $R/gpurun_out/r3w/pmc1.txt 2>&1; cat $R/gpurun_out/r3w/pmc1.txt
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/r3w/pmc2 -- python3 $R/tools/embed_bench.py --layout folded --dist zipf --fields 39 --tag z > $R/gpurun_out/r3w/pmc2.log 2>&1
python3 $R/tools/probes/pmc_table.py $R/gpurun_out/r3w/pmc2 k_apply_main > $R/gpurun_out/r3w/pmc2.txt 2>&1; cat $R/gpurun_out/r3w/pmc2.txt
rocprofv3 --pmc FETCH_SIZE TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/r3w/pmc3 -- python3 $R/tools/embed_bench.py --layout folded --dist zipf --fields 39 --tag z > $R/gpurun_out/r3w/pmc3.log 2>&1
python3 $R/tools/probes/pmc_table.py $R/gpurun_out/r3w/pmc3 k_apply_main > $R/gpurun_out/r3w/pmc3.txt 2>&1; cat $R/gpurun_out/r3w/pmc3.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --output-format csv -d $R/gpurun_out/r3w/pmc4 -- python3 $R/tools/embed_bench.py --layout folded --tag u > $R/gpurun_out/r3w/pmc4.log 2>&1
python3 $R/tools/probes/pmc_table.py $R/gpurun_out/r3w/pmc4 k_apply_main > $R/gpurun_out/r3w/pmc4.txt 2>&1; cat $R/gpurun_out/r3w/pmc4.txt
rm -rf $R/gpurun_out/r3w/pmc1 $R/gpurun_out/r3w/pmc2 $R/gpurun_out/r3w/pmc3 $R/gpurun_out/r3w/pmc4
