"""Times the fused tail launch (ops.tail_fwd_bwd) against the five launches it replaces, B = 16384 (HIP events, 200 calls each)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from mindrec_amd import ops
from test_tail_gpu import _inputs, K2, N2, N3

dev = torch.device("cuda:0")
B, F = 16384, 26
for dt in ("bf16", "f16"):
    x, w2, b2, w3, b3, w5, b5, wide, wb, label = _inputs(dev, dt, B, 1, F=F)
    packed = ops.tail_pack_weights(w2, w3)
    dw5 = torch.empty(N3, device=dev); db4 = torch.empty(N3, device=dev); db5 = torch.empty(1, device=dev)
    db3 = torch.empty(N2, device=dev); db2 = torch.empty(K2, device=dev)
    s3 = torch.empty(N2, device=dev); s2 = torch.empty(K2, device=dev)
    out = {}

    def five():
        y2 = ops.dense_fwd(x, w2, b2, relu=True)
        y3 = ops.dense_fwd(y2, w3, b3, relu=True)
        _, _, _, dz4 = ops.head_fwd_bwd_wide(y3, w5, b5, wide, wb, label, 1024.0 / B, dw5, db4, db5)
        dz3 = ops.dense_bwd_input(dz4, w3, h=y2, db_out=db3)
        ops.dense_bwd_input(dz3, w2, h=x, db_out=db2)

    def one():
        ops.tail_fwd_bwd(x, packed, b2, b3, w5, b5, wide, wb, label, 1024.0 / B, dw5, db4, db5, s3, s2, out=out)

    def tr():
        ops.tail_pack_weights(w2, w3, out=packed)

    for name, fn in (("five launches", five), ("one launch", one), ("weight packing", tr)):
        for _ in range(10):
            fn()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            g.replay()
        b.record(); torch.cuda.synchronize()
        print(f"{dt} {name}: {a.elapsed_time(b) / 200 * 1e3:.1f} us per call (graph of 20, back to back)")
