"""Phase timeline of the fused tail kernel (library built with EXTRA_FLAGS=-DMREC_TAIL_STAMPS OUT=libmrec_stamps.so; MREC_HIP_LIB points at it)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mindrec_amd import ops, _lib
from test_tail_gpu import _inputs, K2, N2, N3
dev = torch.device("cuda:0")
B, F = 16384, 26
x, w2, b2, w3, b3, w5, b5, wide, wb, label = _inputs(dev, "bf16", B, 1, F=F)
packed = ops.tail_pack_weights(w2, w3)
dw5 = torch.empty(N3, device=dev); db4 = torch.empty(N3, device=dev); db5 = torch.empty(1, device=dev)
s3 = torch.empty(N2, device=dev); s2 = torch.empty(K2, device=dev)
out = {}
for _ in range(5):
    ops.tail_fwd_bwd(x, packed, b2, b3, w5, b5, wide, wb, label, 1024.0 / B, dw5, db4, db5, s3, s2, out=out)
torch.cuda.synchronize()
st = np.zeros((512, 8), np.uint64)
l = _lib.lib()
l.mrec_tail_debug_stamps.argtypes = [C.c_void_p]
assert l.mrec_tail_debug_stamps(st.ctypes.data) == 0
khz = C.c_int32(); l.mrec_wall_clock_khz(C.byref(khz))
t = st[:256, :7].astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / (khz.value / 1e3)
names = ["start", "X in LDS", "fwd A done", "fwd B done", "head done", "bwd B done", "end"]
print("phase boundary (us from the first workgroup's start): mean / min / max over 256 workgroups")
for i, n in enumerate(names):
    print(f"  {n:12s} {us[:, i].mean():7.2f} {us[:, i].min():7.2f} {us[:, i].max():7.2f}")
d = np.diff(us, axis=1)
print("phase durations (mean):", " ".join(f"{n}={v:.2f}" for n, v in zip(["load", "fwdA", "fwdB", "head", "bwdB", "bwdA"], d.mean(axis=0))))
