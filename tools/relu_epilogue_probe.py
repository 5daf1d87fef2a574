#!/usr/bin/env python3
"""addmm + relu_ (tuned GEMM, separate ReLU pass) vs torch._addmm_activation (hipBLASLt bias+ReLU epilogue, library
default solution) for the four hidden layers of the Wide&Deep MLP at batch 16384, bf16."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd.wide_deep import enable_tuned_gemms  # noqa: E402

enable_tuned_gemms()
dev = torch.device("cuda:0")
B = 16384
dims = [2080, 1024, 512, 256, 128]


def t(fn, it=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3


for i in range(4):
    x = torch.randn(B, dims[i], device=dev).to(torch.bfloat16)
    W = torch.randn(dims[i], dims[i + 1], device=dev).to(torch.bfloat16)
    b = torch.randn(dims[i + 1], device=dev).to(torch.bfloat16)
    t1 = t(lambda: torch.addmm(b, x, W).relu_())
    t2 = t(lambda: torch._addmm_activation(b, x, W, use_gelu=False))
    same = torch.equal(torch.addmm(b, x, W).relu_(), torch._addmm_activation(b, x, W, use_gelu=False))
    print(f"layer {i} [{B}x{dims[i]}]x[{dims[i]}x{dims[i + 1]}]: addmm+relu_ {t1:6.1f} us   _addmm_activation {t2:6.1f} us   identical output: {same}")
