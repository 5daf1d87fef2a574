import torch
dev = torch.device("cuda:0"); B = 16384
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (K1, N) in [(2080, 1024), (3120, 1024), (1024, 512), (512, 256), (256, 128)]:
    X = torch.randn(B, K1, device=dev, dtype=torch.bfloat16); dY = torch.randn(B, N, device=dev, dtype=torch.bfloat16)
    base = t(lambda: torch.mm(X.t(), dY))
    line = f"dW [{K1}x{B}]x[{B}x{N}]  mm {base:7.1f} us |"
    out32 = torch.empty(K1, N, device=dev)
    for S in (4, 8, 16, 32):
        Xs = X.view(S, B // S, K1); Ys = dY.view(S, B // S, N)
        def f():
            part = torch.bmm(Xs.transpose(1, 2), Ys)            # [S, K1, N] bf16
            torch.sum(part, dim=0, dtype=torch.float32, out=out32)
        tb = t(lambda: torch.bmm(Xs.transpose(1, 2), Ys))
        tt = t(f)
        line += f" S={S}: bmm {tb:6.1f} +sum = {tt:6.1f} |"
    print(line)
    ref = torch.mm(X.float().t(), dY.float())
    part = torch.bmm(X.view(8, B // 8, K1).transpose(1, 2), dY.view(8, B // 8, N)); got = part.float().sum(0)
    print("   rel err splitK8 vs fp32:", float((got - ref).abs().max() / ref.abs().max()), " plain bf16 mm:", float((torch.mm(X.t(), dY).float() - ref).abs().max() / ref.abs().max()))
