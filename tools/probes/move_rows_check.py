import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mindrec_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
for W in (240, 81, 4):
    src = torch.randn(1000, W, generator=g)
    host = src.clone().pin_memory()
    dst = torch.zeros(500, W, device=dev)
    n = 300
    sr = torch.randperm(1000, generator=g)[:n]; dr = torch.randperm(500, generator=g)[:n]
    sr[::7] = -1; dr[3::11] = -1
    nd = torch.tensor([250], dtype=torch.int64, device=dev)
    ops.move_rows_(dst, dr.to(dev), host, sr.to(dev), n_dev=nd)          # host -> device
    ref = torch.zeros(500, W)
    for i in range(250):
        if sr[i] >= 0 and dr[i] >= 0: ref[dr[i]] = src[sr[i]]
    assert torch.equal(dst.cpu(), ref), W
    back = torch.zeros(1000, W).pin_memory()
    ops.move_rows_(back, sr.to(dev), dst, dr.to(dev))                    # device -> host, all n
    torch.cuda.synchronize()
    ref2 = torch.zeros(1000, W)
    for i in range(n):
        if sr[i] >= 0 and dr[i] >= 0: ref2[sr[i]] = ref[dr[i]]
    assert torch.equal(back, ref2), W
print("move_rows ok")
