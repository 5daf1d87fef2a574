# Where does the lookup's time go?  Three builds of the library -- as shipped, without its stores, without its row loads -- timed
# alone on uniform ids x 26 fields and on Zipf ids x 39 fields (run on the GPU box from the repo root).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/gather_abl
mkdir -p $O
# (the two ablated builds come with the snapshot: built in the build container with
#    cd mindrec_amd/csrc && OUT=../../tools/probes/abl/libmrec_abl$a.so EXTRA_FLAGS="-DMREC_GATHER_ABL=$a" bash build.sh   for a in 1 2)
cd $R
for lib in "" $R/tools/probes/abl/libmrec_abl1.so $R/tools/probes/abl/libmrec_abl2.so; do
  echo "== lib: ${lib:-shipped}"
  MREC_HIP_LIB=$lib python tools/embed_bench.py --layout folded --tag u26 2>/dev/null | grep "lookup"
  MREC_HIP_LIB=$lib python tools/embed_bench.py --layout folded --dist zipf --fields 39 --tag z39 2>/dev/null | grep "lookup"
  MREC_HIP_LIB=$lib python tools/embed_bench.py --layout folded --dist zipf --fields 26 --tag z26 2>/dev/null | grep "lookup"
done > $O/ablation.txt
cat $O/ablation.txt
