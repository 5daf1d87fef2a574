"""Is a forward GEMM with a TRANSPOSED weight (both operands K-contiguous) faster than the one that reads W [K, N] as stored?
The input-gradient kernel without a mask IS that GEMM: dx[M, Kout] = dy[M, Kred] . w[Kout, Kred]^T."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd import ops
dev = torch.device("cuda:0")
M = 16384
def timeit(fn, n=20):
    for _ in range(5): fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): g.replay()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / (5 * n) * 1e3
for K, N in ((2080, 1024), (1024, 512)):
    x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
    w = (torch.randn(K, N, device=dev) * 0.05).to(torch.bfloat16)
    wt = w.t().contiguous()
    b = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    t_tn = timeit(lambda: ops.dense_fwd(x, w, b, relu=True, out=y))
    t_nt = timeit(lambda: ops.dense_bwd_input(x, wt, out=y))
    print(f"K={K} N={N}: forward as stored {t_tn:.1f} us, with transposed weight (NT) {t_nt:.1f} us")
