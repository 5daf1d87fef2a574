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
        # build -> verify on this first batch -> (distributed cells: all ranks agree) -> commit; a refusal leaves the cell untouched
        low = lowering.lower_train_step(cell, batch)
        d["_lowered"] = low or False
        if not low:
            # a GRAPH_MODE train cell that stays on the interpreter is worth a WARNING, once: the user would otherwise benchmark
            # primitive-by-primitive execution without being told
            from . import log
            log.warning("GRAPH_MODE: %s is NOT lowered onto the fused engine and runs primitive by primitive: %s", type(cell).__name__,
                        d.get("_lowering_refused", "not lowered"))
    return low or None
