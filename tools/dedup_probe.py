import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mindrec_amd import ops
dev = torch.device("cuda:0")
B = 16384
def t(name, ids):
    ids = ids.to(dev)
    for _ in range(3): ops.unique(ids)
    torch.cuda.synchronize()
    evs = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); d = ops.unique(ids); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    ts = sorted(x.elapsed_time(y) for x, y in evs)
    print(f"{name:40s} n={ids.numel():7d} U={d.U:7d} dedup median {ts[10]*1e3:8.1f} us")
rng = np.random.default_rng(0)
V = 200_000_000
u26 = torch.from_numpy(rng.integers(0, V, size=(B, 26)).astype(np.int32))
t("uniform F=26", u26)
u39 = rng.integers(13, V, size=(B, 39)).astype(np.int32)
t("uniform F=39", torch.from_numpy(u39))
h = u39.copy(); h[:, :13] = np.arange(13)
t("uniform + 13 constant ids", torch.from_numpy(h))
h1 = u39.copy(); h1[:, 0] = 7
t("uniform + 1 constant id", torch.from_numpy(h1))
z = np.minimum(rng.zipf(1.05, size=(B, 39)) - 1, 5_000_000) + 13 + 5_000_000 * np.arange(39)[None, :]
t("zipf(1.05) per slot, no constants", torch.from_numpy(z.astype(np.int32)))
z2 = np.minimum(rng.zipf(1.5, size=(B, 39)) - 1, 5_000_000) + 13 + 5_000_000 * np.arange(39)[None, :]
t("zipf(1.5) per slot", torch.from_numpy(z2.astype(np.int32)))
s = np.full((B, 39), 5, np.int32)
t("all same", torch.from_numpy(s))
r = (np.arange(B * 39) % 1000).reshape(B, 39).astype(np.int32)
t("1000 ids round-robin", torch.from_numpy(r))
