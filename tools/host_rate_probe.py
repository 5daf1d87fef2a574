#!/usr/bin/env python3
"""Is the Wide&Deep step host-bound?  Host time to ISSUE one step (no sync) vs device time per step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine, synthetic_batch  # noqa: E402

dev = torch.device("cuda:0")
graph = "--no-graph" not in sys.argv
nb = 4 if "--rotate" in sys.argv else 1
cfg = WideDeepConfig(field_size=26, graphs="step" if graph else "none")
eng = WideDeepEngine(cfg, dev)
batches = [synthetic_batch(cfg, dev, "uniform", 1000 + i) for i in range(nb)]
for i in range(5):
    eng.train_step(*batches[i % nb])
if "--timers" in sys.argv:
    eng.timers = {}
torch.cuda.synchronize()
N = 40
host = []
t0 = time.perf_counter()
for i in range(N):
    a = time.perf_counter()
    eng.train_step(*batches[i % nb])
    host.append(time.perf_counter() - a)
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
host.sort()
print(" ".join(sys.argv[1:]) or "(default)")
print(f"host issue time/step: median {host[N // 2] * 1e3:.3f} ms, min {host[0] * 1e3:.3f} ms; all issued after {t_issue * 1e3:.1f} ms; "
      f"device done after {t_all * 1e3:.1f} ms  ({t_all / N * 1e3:.3f} ms/step)")
