import time, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
B = 16384
dims = [2080, 1024, 512, 256, 128, 1]
Ws = [torch.randn(dims[i], dims[i+1], device=dev, requires_grad=True) * 0.01 for i in range(5)]
Ws = [w.detach().requires_grad_(True) for w in Ws]
bs = [torch.zeros(dims[i+1], device=dev, requires_grad=True) for i in range(5)]
x = torch.randn(B, dims[0], device=dev)
label = (torch.rand(B, 1, device=dev) < 0.25).float()

def run(variant, iters=20):
    def step():
        xx = x.detach().requires_grad_(True)
        if variant == "addmm_bf16":
            h = xx.to(torch.bfloat16)
            for i in range(5):
                h = torch.addmm(bs[i].to(torch.bfloat16), h, Ws[i].to(torch.bfloat16))
                if i < 4: h = torch.relu(h)
            h = h.float()
        elif variant == "autocast":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                h = xx
                for i in range(5):
                    h = F.linear(h, Ws[i].t(), bs[i])
                    if i < 4: h = torch.relu(h)
            h = h.float()
        elif variant == "fp32":
            h = xx
            for i in range(5):
                h = torch.addmm(bs[i], h, Ws[i])
                if i < 4: h = torch.relu(h)
        loss = F.binary_cross_entropy_with_logits(h, label)
        (loss * 1024).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters): step()
    th = time.perf_counter() - t
    torch.cuda.synchronize()
    tt = time.perf_counter() - t
    print(f"{variant:12s} host-issue {th/iters*1e3:8.3f} ms/iter   wall {tt/iters*1e3:8.3f} ms/iter")

for v in ["addmm_bf16", "autocast", "fp32", "addmm_bf16"]:
    run(v)
# single op host cost
a = torch.randn(16384, 2080, device=dev, dtype=torch.bfloat16); w = torch.randn(2080, 1024, device=dev, dtype=torch.bfloat16)
for name, fn in [("mm_bf16", lambda: torch.mm(a, w)), ("relu", lambda: torch.relu(a)), ("empty", lambda: torch.empty(16384*2080, device=dev))]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(50): fn()
    th = time.perf_counter() - t; torch.cuda.synchronize(); tt = time.perf_counter() - t
    print(f"{name:12s} host {th/50*1e6:8.1f} us/op   wall {tt/50*1e6:8.1f} us/op")
import os; print("cpus", os.cpu_count(), "threads", torch.get_num_threads())
