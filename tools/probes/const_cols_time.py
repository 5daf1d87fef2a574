import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, time
from mindrec_amd import ops
from mindrec_amd.wide_deep import WideDeepConfig, synthetic_batch
dev=torch.device("cuda:0")
for dist,F in (("zipf",39),("uniform",39),("uniform",26),("zipf",26)):
    cfg=WideDeepConfig(vocab_size=200_000_000, emb_dim=80, field_size=F, batch_size=16384)
    ids,_,_=synthetic_batch(cfg, dev, dist, seed=1)
    st=ops.const_cols_state(dev)
    for mc in (None, 2048):
        for _ in range(5): ops.const_cols_detect(ids, cfg.vocab_size, st, min_count=mc)
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): ops.const_cols_detect(ids, cfg.vocab_size, st, min_count=mc)
        e1.record(); torch.cuda.synchronize()
        print(dist, F, "min_count", mc, "%.1f us per call" % (e0.elapsed_time(e1)*1000/50), "mask bits", bin(ops.const_cols_mask(st)).count("1"))
