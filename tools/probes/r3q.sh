mkdir -p gpurun_out/r3q
MREC_HIP_LIB=mindrec_amd/csrc/libmrec_aw8_ab2_gp8_w4.so python bench.py --no-cpu-baseline --no-zipf39 --steps 20 > gpurun_out/r3q/v.out 2> gpurun_out/r3q/v.err; tail -6 gpurun_out/r3q/v.err; cut -c1-100 gpurun_out/r3q/v.out
