"""The sparse apply's finishing pass (runs of duplicates that cross windows of the sorted index) deferred into the dense net's
Adam launch (mrec_sparse_lazy_adam_wide_defer + mrec_dense_adam_slabs_finish_f32): bit-identical to the separate launches --
on the tables, the optimizer state, the wide records and the dense buffers -- for ids with many long runs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
@pytest.mark.parametrize("gdt", [torch.float16, torch.bfloat16])
def test_deferred_finishing_pass_is_bit_identical(dev, idt, gdt):
    from mindrec_amd import ops
    rng = np.random.default_rng(3)
    V, D, B, F = 5000, 80, 512, 39
    ld = 256
    ids = np.minimum(rng.zipf(1.1, size=(B, F)) + 12, V - 1)
    ids[:, :13] = np.arange(13)                                   # the 13 constant dense-field ids: runs of 512 copies
    tid = torch.from_numpy(ids).to(dev, idt)
    wts = torch.from_numpy(rng.random((B, F)).astype(np.float32)).to(dev)
    g = torch.from_numpy(rng.standard_normal((B * F, D)).astype(np.float32)).to(dev, gdt)
    gw = torch.from_numpy(rng.standard_normal(B).astype(np.float32)).to(dev)
    n_dense = 4096
    slabs = torch.from_numpy(rng.standard_normal((3, 1024)).astype(np.float32)).to(dev)
    gd = torch.from_numpy(rng.standard_normal(n_dense).astype(np.float32)).to(dev)

    def fresh():
        st = torch.from_numpy(rng0.standard_normal((V, ld)).astype(np.float32) * 0.01).to(dev)
        st[:, D + 1] = 1.0                                          # FTRL accum
        st[:, 2 * D + 4:3 * D + 4] = st[:, 2 * D + 4:3 * D + 4].abs()   # Adam v >= 0
        dense = [torch.from_numpy(a.copy()).to(dev) for a in dn]
        dense[2].abs_()
        return st, dense

    outs = []
    for defer in (False, True):
        rng0 = np.random.default_rng(11)
        dn = [rng0.standard_normal(n_dense).astype(np.float32) * 0.01 for _ in range(3)]
        st, (p, m, v) = fresh()
        shadow = torch.zeros(n_dense, dtype=torch.float16, device=dev)
        plan = ops.sparse_plan(tid)
        kw = dict(lr=3.5e-4, beta1_power=0.9, beta2_power=0.999, grad_scale=1 / 1024)
        fin = ops.sparse_lazy_adam_wide_(st[:, :D], st[:, D + 4:2 * D + 4], st[:, 2 * D + 4:3 * D + 4], plan, g, wts, gw, F, D, defer=defer, **kw)
        assert (fin is not None) == defer
        ops.dense_adam_slabs_(p, m, v, gd, [(1024, slabs)], shadow16=shadow, ftrl1=(n_dense - 4, 5e-2, 1e-8, 1e-8, -0.5), finish=fin, **kw)
        torch.cuda.synchronize()
        outs.append([t.cpu().numpy() for t in (st, p, m, v, shadow.view(torch.int16))])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    assert not np.array_equal(outs[0][0][:13, :D], np.zeros((13, D)))
