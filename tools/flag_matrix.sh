for f in "--no-graph-front" "--no-graph-mlp" "--mlp-dtype fp32" "--split-state" "--no-overlap-plan" "--late-wide on" "--overlap-dw0" "--vocab 3000000 --dynamic-embedding" "--vocab 4000000 --host-cache-rows 1000000 --dist zipf" "--batch 8192" "--emb-dim 64 --vocab 50000000" "--fields 39 --dist zipf --steps 10 --warmup 0"; do
  python bench.py --no-cpu-baseline $f > gpurun_out/fl.log 2> gpurun_out/fl.err || { echo "FAILED [$f]"; tail -3 gpurun_out/fl.err; continue; }
  echo "[$f]" $(tail -1 gpurun_out/fl.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['config']['hip_graphs'])")
done
