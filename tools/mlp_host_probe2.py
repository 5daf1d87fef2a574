import time, torch, torch.nn.functional as F
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
B = 16384
dims = [2080, 1024, 512, 256, 128, 1]
Ws = [(torch.randn(dims[i], dims[i+1], device=dev) * 0.01).requires_grad_(True) for i in range(5)]
bs = [torch.zeros(dims[i+1], device=dev, requires_grad=True) for i in range(5)]
x = torch.randn(B, dims[0], device=dev)
label = (torch.rand(B, 1, device=dev) < 0.25).float()
def step():
    xx = x.detach().requires_grad_(True)
    h = xx.to(torch.bfloat16)
    for i in range(5):
        h = torch.addmm(bs[i].to(torch.bfloat16), h, Ws[i].to(torch.bfloat16))
        if i < 4: h = torch.relu(h)
    h = h.float()
    loss = F.binary_cross_entropy_with_logits(h, label)
    (loss * 1024).backward()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(5): step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))
# time individual bf16 gemm shapes (host cost)
def t(name, fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0; torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"{name:40s} host {th/n*1e6:9.1f} us  wall {tt/n*1e6:9.1f} us")
bf = torch.bfloat16
for (m, k, n_) in [(16384, 2080, 1024), (16384, 128, 1), (16384, 1, 128), (128, 16384, 1), (2080, 16384, 1024), (16384, 1024, 2080)]:
    a = torch.randn(m, k, device=dev, dtype=bf); b = torch.randn(k, n_, device=dev, dtype=bf)
    t(f"mm bf16 [{m},{k}]x[{k},{n_}]", lambda: torch.mm(a, b))
    at = torch.randn(k, m, device=dev, dtype=bf)
    t(f"mm bf16 T [{k},{m}]^T x [{k},{n_}]", lambda: torch.mm(at.t(), b))
