#!/usr/bin/env python3
"""Which operand layouts do the GEMM libraries like best for the three big products of the first MLP layer
(B = 16384, K = 2080, N = 1024, bf16)?  Each variant is tuned with TunableOp (best solution of hipBLASLt / rocBLAS),
then timed with HIP events."""
import os
import sys

os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
os.environ["PYTORCH_TUNABLEOP_FILENAME"] = "gpurun_out/layout_probe.csv"
os.environ["PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS"] = "60"
import torch  # noqa: E402

dev = torch.device("cuda:0")
B, K, N = 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 2080, 1024
bf = torch.bfloat16
x = torch.randn(B, K, device=dev).to(bf)
W = torch.randn(K, N, device=dev).to(bf)          # as stored: [K, N]
WT = W.t().contiguous()                            # [N, K]
dh = torch.randn(B, N, device=dev).to(bf)
xT = x.t().contiguous()                            # [K, B]
dhT = dh.t().contiguous()                          # [N, B]
bias = torch.randn(N, device=dev).to(bf)


def t(fn, it=30):
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


fl = 2.0 * B * K * N
for name, fn in [
    ("fwd  addmm(bias, x, W)            [NN]", lambda: torch.addmm(bias, x, W)),
    ("fwd  addmm(bias, x, WT.t())       [NT]", lambda: torch.addmm(bias, x, WT.t())),
    ("dX   mm(dh, W.t())                [NT]", lambda: torch.mm(dh, W.t())),
    ("dX   mm(dh, WT)                   [NN]", lambda: torch.mm(dh, WT)),
    ("dW   mm(x.t(), dh)                [TN]", lambda: torch.mm(x.t(), dh)),
    ("dW   mm(xT, dh)                   [NN]", lambda: torch.mm(xT, dh)),
    ("dW^T mm(dh.t(), x)                [TN]", lambda: torch.mm(dh.t(), x)),
    ("dW^T mm(dhT, x)                   [NN]", lambda: torch.mm(dhT, x)),
    ("dW   bmm split-K 8 (x^T chunks . dh chunks)", lambda: torch.bmm(x.view(8, B // 8, K).transpose(1, 2), dh.view(8, B // 8, N))),
    ("dW^T bmm split-K 8 (dh^T chunks . x chunks)", lambda: torch.bmm(dh.view(8, B // 8, N).transpose(1, 2), x.view(8, B // 8, K))),
    ("dW   bmm split-K 4", lambda: torch.bmm(x.view(4, B // 4, K).transpose(1, 2), dh.view(4, B // 4, N))),
    ("dW   bmm split-K 16", lambda: torch.bmm(x.view(16, B // 16, K).transpose(1, 2), dh.view(16, B // 16, N))),
]:
    us = t(fn)
    print(f"{name:48s} {us:7.1f} us  {fl / us / 1e9:6.2f} PFLOP/s" .replace("PFLOP/s", "TFLOP/s x1e-3") if False else f"{name:48s} {us:7.1f} us  {fl / us / 1e6 / 1e3:6.3f} PFLOP/s")
