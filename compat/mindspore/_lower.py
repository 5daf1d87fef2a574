"""GRAPH_MODE's "compile" on an MI355X: the first top-level call of a training cell asks mindrec_amd.lowering whether the cell is
a train step it can run as the fused engine (one HIP graph per step); the answer is cached on the cell."""
import torch

from . import context


def lowered(cell, batch=None):
    """-> the LoweredStep of `cell`, or None (not GRAPH_MODE / not the GPU target / not a recognised train step)."""
    if context.get_context("mode") != context.GRAPH_MODE or context.get_context("device_target") != "GPU" or context._host_tensors:
        return None
    d = cell.__dict__
    low = d.get("_lowered")
    if low is None:
        if not any(hasattr(c, "loss_scale") and hasattr(c, "parameters") for c in cell.cells()):
            d["_lowered"] = False                    # no optimizer among its children: not a train step
            return None
        try:
            from mindrec_amd import lowering
        except ImportError:
            d["_lowered"] = False
            return None
        low = lowering.lower_train_step(cell)
        if low is not None and batch is not None and len(batch) >= 2 and all(isinstance(t, torch.Tensor) for t in batch[:2]):
            try:
                low.verify(batch[0], batch[1])
            except lowering.LoweringRefused as e:
                raise RuntimeError(f"lowering of {type(cell).__name__} failed its verification after the parameters were "
                                   f"re-bound: {e}") from e
        d["_lowered"] = low or False
        if not low:
            from . import log
            log.info("GRAPH_MODE: %s stays on the primitive-by-primitive path: %s", type(cell).__name__,
                     d.get("_lowering_refused", "not lowered"))
    return low or None
