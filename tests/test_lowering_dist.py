"""CPU, gloo, world 2: the lowering's DISTRIBUTION decision (mindrec_amd/lowering.py: dist_plan, the agreement all-reduce) is the
same on every rank, and a cell that reduces its gradients over more than one rank is never handed a one-rank engine.  The GPU
twin -- the decisions carried out, against the reference's own data-parallel fixtures -- is tests/test_lowered_dist_gpu.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "compat")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import mindspore
    import _ms_cpu_kernels
    from mindspore import context
    from mindspore.communication.management import init
    mindspore._kernels._install(_ms_cpu_kernels)
    context.set_context(mode=context.GRAPH_MODE, device_target="CPU")
    init("gloo")
    context.set_auto_parallel_context(parallel_mode=context.ParallelMode.DATA_PARALLEL, gradients_mean=True, device_num=world)
    import _ms_models
    import _ref_fixtures as RF
    from mindrec_amd import lowering
    got = {}
    # (1) row gradients + reducers: the plan is "shard" on every rank (on host tensors the build then stops at the device check)
    z, cfg, comp = RF.load("ref_wd_sparse")
    step, _ = _ms_models.wide_deep_from_fixture(z, cfg, comp, reduce=True)
    plan = lowering.dist_plan(step)
    got["sparse_plan"] = [plan.world, plan.rank, plan.reduces, plan.shard]
    assert lowering.lower_train_step(step) is None
    got["sparse_refused"] = step._lowering_refused
    # (2) dense table gradients + reducers: refused by name, never a one-rank engine
    z, cfg, comp = RF.load("ref_wd_dense")
    step, _ = _ms_models.wide_deep_from_fixture(z, cfg, comp, reduce=True)
    assert lowering.lower_train_step(step) is None
    got["dense_refused"] = step._lowering_refused
    # (3) a reducer that does not average over all ranks: refused
    step, _ = _ms_models.wide_deep_from_fixture(z, cfg, comp, reduce=True)
    step.reduce_d.degree = 1
    assert lowering.lower_train_step(step) is None
    got["degree_refused"] = step._lowering_refused
    # (4) data-parallel context, reducer_flag, but no reducer to read: refused
    step, _ = _ms_models.wide_deep_from_fixture(z, cfg, comp, reduce=False)
    step.reducer_flag = True
    assert lowering.lower_train_step(step) is None
    got["flag_refused"] = step._lowering_refused
    # (5) one rank would lower, the other refuses: BOTH end up refused (the agreement all-reduce), the would-be engine is dropped
    z, cfg, comp = RF.load("ref_wd_sparse")
    step, _ = _ms_models.wide_deep_from_fixture(z, cfg, comp, reduce=True)
    dropped = []

    class _Fake:
        def discard(self):
            dropped.append(True)

    def fake(cell, plan):
        if rank == 0:
            return _Fake()
        raise lowering.LoweringRefused("rank 1 says no")
    orig = lowering._lower_wide_deep
    lowering._lower_wide_deep = fake
    try:
        assert lowering.lower_train_step(step) is None
    finally:
        lowering._lower_wide_deep = orig
    got["split_refused"], got["split_dropped"] = step._lowering_refused, bool(dropped)
    # (6) Deep&Cross under a reducer: refused (a one-rank engine)
    z, cfg, comp = RF.load("ref_dcn")
    step, _ = _ms_models.deep_cross_from_fixture(z, cfg, comp)
    from mindspore.nn.wrap.grad_reducer import DistributedGradReducer
    step.reducer = DistributedGradReducer(step.weights, True, world)
    assert lowering.lower_train_step(step) is None
    got["dcn_refused"] = step._lowering_refused
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([got], dtype=object), allow_pickle=True)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_the_distribution_decision_is_the_same_on_every_rank(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(tmp_path / f"rank{k}.npy", allow_pickle=True)[0] for k in range(2))
    assert a["sparse_plan"] == [2, 0, True, True] and b["sparse_plan"] == [2, 1, True, True]
    assert a["sparse_refused"] == b["sparse_refused"] == "parameters are not on an MI355X"
    assert a["dense_refused"] == b["dense_refused"] and "dense table gradients" in a["dense_refused"]
    assert a["degree_refused"] == b["degree_refused"] and "only mean over all ranks" in a["degree_refused"]
    assert a["flag_refused"] == b["flag_refused"] and "without a DistributedGradReducer" in a["flag_refused"]
    assert a["split_refused"] == "another rank refused the lowering" and a["split_dropped"]
    assert b["split_refused"] == "rank 1 says no" and not b["split_dropped"]
    assert a["dcn_refused"] == b["dcn_refused"] and "one-rank engine" in a["dcn_refused"]


def test_one_process_plan_is_single_rank():
    """No process group: the plan is the one-rank one whatever reducers the cell owns (they are identities at world 1)."""
    for p in (ROOT, os.path.join(ROOT, "compat")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from mindrec_amd import lowering

    class _C:
        def cells_and_names(self):
            return [("", self)]
    plan = lowering.dist_plan(_C())
    assert (plan.world, plan.rank, plan.reduces, plan.shard) == (1, 0, False, False)
