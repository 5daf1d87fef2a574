"""Generates the committed golden fixtures from the CPU oracle.

SOURCE OF THESE VECTORS: our restatement (oracle/mrec_oracle.c), NOT MindSpore -- the reference's
own tests hold no vectors for this path and MindSpore cannot be imported here.
They pin (a) the bit stream of the table initialiser shared by CPU and GPU, (b) one Wide&Deep
embedding step at BASELINE config-1 shape (dim 16, 39 fields) and (c) the Dropout mask function, so that a change to either side's
arithmetic is caught even when oracle and kernels are edited together.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rows = [0, 1, 12, 199_999_999, 2**40 + 7, -5]
    D, sigma, seed = 8, 0.01, 1000
    bits = O.normal_rows(seed, rows, D, sigma).view(np.uint32).ravel().tolist()
    json.dump({"seed": seed, "rows": rows, "D": D, "sigma": sigma, "bits": bits},
              open(os.path.join(HERE, "normal_seed1000.json"), "w"))

    # (c) the Dropout mask function (csrc/mrec_dropout.h / oracle.dropout_mask): kept-bits of a few small masks
    cases = [dict(M=5, W=16, seed=1004, step=0, layer=0, keep=0.5, row0=0), dict(M=3, W=8, seed=1004, step=7, layer=4, keep=0.8, row0=16384),
             dict(M=2, W=12, seed=2 ** 63 + 11, step=100000, layer=15, keep=0.25, row0=3)]
    for c in cases:
        mk = O.dropout_mask(c["M"], c["W"], c["seed"], c["step"], c["layer"], c["keep"], c["row0"])
        c["kept"] = "".join("1" if x > 0 else "0" for x in mk.ravel())
    json.dump(cases, open(os.path.join(HERE, "dropout_masks.json"), "w"))

    rng = np.random.default_rng(1000)          # set_seed(1000), train_and_eval_distribute.py:72
    V, Dm, B, F = 20000, 16, 64, 39
    ids = np.minimum(rng.zipf(1.05, size=(B, F)) + 12, V - 1).astype(np.int32)
    ids[:, :13] = np.arange(13, dtype=np.int32)      # dense fields -> constant ids 0..12 (process_data.py:138-147)
    wts = np.ones((B, F), np.float32)
    wts[:, :13] = rng.random((B, 13)).astype(np.float32)
    g = (rng.standard_normal((B, F, Dm)) * 1024 * 1e-3).astype(np.float32)
    p = O.fill_normal(seed, V, Dm, 0.01)
    m = np.zeros_like(p); v = np.zeros_like(p)
    emb = O.gather_rows(p, ids, wts)
    O.sparse_lazy_adam(p, m, v, ids, g, wts, lr=3.5e-4, eps=1e-8, b1_pow=0.9, b2_pow=0.999, grad_scale=1 / 1024)
    touched = np.unique(ids)
    np.savez_compressed(os.path.join(HERE, "wd_step_small.npz"), seed=seed, V=V, D=Dm, ids=ids, wts=wts, g=g, emb=emb,
                        p_touched=p[touched], m_touched=m[touched], v_touched=v[touched])


if __name__ == "__main__":
    main()
