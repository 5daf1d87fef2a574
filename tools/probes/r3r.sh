mkdir -p gpurun_out/r3r
for lib in "" mindrec_amd/csrc/libmrec_gf32_e1.so mindrec_amd/csrc/libmrec_gf32_e2.so mindrec_amd/csrc/libmrec_gf32_e3.so; do echo "lib=${lib:-default}"; MREC_HIP_LIB=$lib python tools/dcn_bench.py 2>/dev/null | tail -3; done > gpurun_out/r3r/exp.txt 2>&1; cat gpurun_out/r3r/exp.txt
