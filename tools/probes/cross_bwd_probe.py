#!/usr/bin/env python3
"""Cross-stack backward at configs[2] width over a range of batch sizes (fixed part vs per-row part), HIP events; the kernel
variant comes from MREC_CROSS_BWD_VAR.  Results are compared with the ones variant 0 left under /tmp (or argv[1])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
D, L = 1170, 6
var = os.environ.get("MREC_CROSS_BWD_VAR", "0")
out_dir = sys.argv[1] if len(sys.argv) > 1 else "/tmp"
g = torch.Generator(device="cpu").manual_seed(5)
line = []
for B in [int(v) for v in os.environ.get("CROSS_PROBE_B", "2048,16384,32768").split(",")]:
    x0 = (torch.randn(B, D, generator=g) * 0.5).to(dev)
    w = (torch.randn(L, D, generator=g) / D ** 0.5).to(dev)
    b = (torch.randn(L, D, generator=g) * 0.1).to(dev)
    dy = torch.randn(B, D, generator=g).to(dev)
    for _ in range(5):
        r = ops.cross_layers_bwd(x0, w, b, dy)
    for _ in range(55):                      # the forward as well, for the kernel trace
        ops.cross_layers(x0, w, b)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        r = ops.cross_layers_bwd(x0, w, b, dy)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 50 * 1e3
    ref_path = os.path.join(out_dir, f"cross_bwd_ref_{B}.pt")
    note = ""
    if var == "0":
        torch.save([t.cpu() for t in r], ref_path)
    elif os.path.exists(ref_path):
        ref = torch.load(ref_path)
        errs = [float((a.cpu() - c).abs().max() / c.abs().max()) for a, c in zip(r, ref)]
        note = " max rel err vs variant 0: " + "/".join(f"{e:.1e}" for e in errs)
    line.append(f"variant {var}  B = {B:6d}: {us:7.1f} us   {3 * B * D * 4 / us / 1e6:6.2f} TB/s{note}")
print("\n".join(line))
