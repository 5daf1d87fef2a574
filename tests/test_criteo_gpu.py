"""The real-data path end to end on the GPU: Criteo-format TSV lines -> StatsDict (the reference's id /
weight encoding, datasets/criteo_1tb/process_data.py:43-163) -> records of `line_per_sample` samples (:203-283) -> the
rank-sharded record reader (models/wide_deep/src/datasets.py:274-325) -> WideDeepEngine.train_step -> AUC on held-out records
(src/metrics.py:37-52).  The oracle-side engine is fed the same batches: tables, losses and the AUC must agree."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _tsv(n, seed):
    """Criteo-format lines with a planted signal: the label depends on three of the categorical columns and one integer column."""
    rng = np.random.default_rng(seed)
    cat_w = {j: rng.normal(0, 1.2, size=60) for j in (0, 5, 17)}
    lines = []
    for _ in range(n):
        dense = ["" if rng.random() < 0.1 else str(int(rng.integers(0, 50))) for _ in range(13)]
        cv = [int(min(rng.zipf(1.3), 59)) for _ in range(26)]
        cats = ["%08x" % (v * 7919 + j) if rng.random() > 0.03 else "" for j, v in enumerate(cv)]
        z = sum(cat_w[j][cv[j]] for j in cat_w) + (0.02 * float(dense[3]) if dense[3] else 0.0) - 0.5
        lab = int(rng.random() < 1.0 / (1.0 + np.exp(-z)))
        lines.append("\t".join([str(lab)] + dense + cats))
    return lines


@pytest.mark.timeout(900)
def test_tsv_to_records_to_engine_to_auc(dev, oracle, tmp_path):
    from _oracle_engine import OracleWideDeepEngine
    from mindrec_amd.criteo import RecordDataset, StatsDict, write_records
    from mindrec_amd.wide_deep import WideDeepConfig, WideDeepEngine
    from mindrec_amd.wide_deep_run import AUCMetric, WideDeepRunner
    lps = 100                                             # samples per record (the reference packs 1000)
    train, test = _tsv(6000, 1), _tsv(1500, 2)
    st = StatsDict()
    st.update(train[:3000]); st.update(train[3000:])      # chunked first pass, as over the day files
    st.finalize()
    assert write_records(str(tmp_path), "train", *st.encode(train), records_per_file=16, line_per_sample=lps) == 60
    assert write_records(str(tmp_path), "test", *st.encode(test), records_per_file=16, line_per_sample=lps) == 15
    B = 1000
    cfg = WideDeepConfig(vocab_size=st.vocab_size, emb_dim=16, field_size=39, batch_size=B, deep_layer_dim=[64, 32], mlp_dtype="fp32",
                         adam_lr=3e-3)
    g, o = WideDeepEngine(cfg, dev), OracleWideDeepEngine(cfg, "cpu")
    as_t = lambda a: torch.from_numpy(a)      # noqa: E731
    steps = 0
    for rank in (0, 1):                       # the two ranks' shards of the records, one after the other: 30 records = 3 batches each
        ds = RecordDataset(str(tmp_path), train_mode=True, batch_size=B, line_per_sample=lps, rank_size=2, rank_id=rank, seed=3, to_device=as_t)
        assert ds.get_dataset_size() == 3
        for epoch in range(8):
            for ids, wts, label in ds:
                assert ids.shape == (B, 39) and ids.dtype == torch.int32 and int(ids.max()) < st.vocab_size
                lg = float(g.train_step(ids.to(dev), wts.to(dev), label.to(dev)))
                lo = float(o.train_step(ids, wts, label))
                assert abs(lg - lo) <= 1e-4 * abs(lo)
                steps += 1
            ds.reset()
    assert steps == 48
    for name in ("deep", "wide", "deep_m"):
        a, b = getattr(g, name).cpu().numpy(), getattr(o, name).numpy()
        # 48 free-running fp32 steps on two stacks (GEMM summation orders differ): Adam moves an element by ~lr per step
        # whatever its gradient's size, so an element whose gradient is noise moves by a step's fraction either way:
        # 99 % of the elements within 0.2 lr, none further apart than one step of the optimizer
        d = np.abs(a - b)
        q = np.quantile(d, [0.5, 0.9, 0.99])
        print(f"  {name}: |diff| median {q[0]:.2e}, 90 % {q[1]:.2e}, 99 % {q[2]:.2e}, max {d.max():.2e} (adam lr {cfg.adam_lr}, ftrl lr {cfg.ftrl_lr})")
        assert q[2] <= 0.2 * cfg.adam_lr and d.max() <= max(cfg.adam_lr, cfg.ftrl_lr), name
    ev_g = RecordDataset(str(tmp_path), train_mode=False, batch_size=500, line_per_sample=lps, to_device=lambda a: torch.from_numpy(a).to(dev))
    ev_o = RecordDataset(str(tmp_path), train_mode=False, batch_size=500, line_per_sample=lps, to_device=as_t)
    auc_g = WideDeepRunner(g, {"auc": AUCMetric()}).eval(ev_g)["auc"]
    auc_o = WideDeepRunner(o, {"auc": AUCMetric()}).eval(ev_o)["auc"]
    print(f"  held-out AUC over {ev_g.get_dataset_size() * 500} samples: gpu {auc_g:.5f}, oracle {auc_o:.5f}")
    assert auc_g > 0.6 and abs(auc_g - auc_o) < 1e-3
