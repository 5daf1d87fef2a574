"""Times the three-part bf16 fp32 GEMM (ops.x3_gemm) against the exact-fp32 MFMA kernel (ops.dense32_*) at Deep&Cross's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd import ops  # noqa: E402


def t(fn, n=40):
    for _ in range(10):          # (clocks settle: a 10-call sample right behind the fp32-MFMA kernel read 339 us where rocprof shows 228)
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


dev = torch.device("cuda:0")
M = 16384
for K, N in ((1170, 1024), (1024, 1024)):
    x = torch.randn(M, K, device=dev) * 0.1
    w = torch.randn(K, N, device=dev) * 0.05
    dy = torch.randn(M, N, device=dev) * 0.01
    xp, wp, dyp = ops.x3_split(x), ops.x3_split(w), ops.x3_split(dy)
    y = torch.empty(M, N, device=dev)
    dx = torch.empty(M, K, device=dev)
    S = int(os.environ.get('X3_S', 0)) or ops.x3_slabs(M, ops.dense32_bwd_weight_slabs(M, K, N))
    dw = torch.empty(S, K, N, device=dev)
    fl = 2 * M * K * N
    r = {"split x": t(lambda: ops.x3_split(x, out=xp)),
         "fwd x3": t(lambda: ops.x3_gemm(0, xp, wp, M, K, N, y)), "fwd f32": t(lambda: ops.dense32_fwd(x, w, None, relu=False, out=y)),
         "dgrad x3 (split tile)": t(lambda: ops.x3_dgrad(dyp, wp, M, K, N, dx)),
         "dgrad x3": t(lambda: ops.x3_gemm(1, dyp, wp, M, K, N, dx)), "dgrad f32": t(lambda: ops.dense32_bwd_input(dy, w, out=dx)),
         "wgrad x3": t(lambda: ops.x3_gemm(2, xp, dyp, M, K, N, dw, S=S)), "wgrad f32": t(lambda: ops.dense32_bwd_weight(x, dy, dw)),
         "bias_relu+parts": t(lambda: ops.x3_bias_relu_(y, torch.zeros(N, device=dev), True, dyp))}
    print(f"M {M} K {K} N {N} (S {S}): " + ", ".join(f"{k} {v:.0f} us" + (f" ({fl / v / 1e6:.0f} TF/s)" if "x3" in k or "f32" in k else "") for k, v in r.items()))
