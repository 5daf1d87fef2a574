import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from mindrec_amd import ops
dev = torch.device("cuda:0")
M, K, N = 16384, 1170, 1024
w = torch.randn(K, N, device=dev) * 0.05
dy = torch.randn(M, N, device=dev) * 0.01
wp, dyp = ops.x3_split(w), ops.x3_split(dy)
dx = torch.empty(M, K, device=dev)
for name, fn in (("x3_dgrad (split)", lambda: ops.x3_dgrad(dyp, wp, M, K, N, dx)), ("x3_gemm form 1", lambda: ops.x3_gemm(1, dyp, wp, M, K, N, dx))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: host issue {1e6 * (t1 - t0) / 20:.1f} us per call, total {1e6 * (t2 - t0) / 20:.1f} us per call")
