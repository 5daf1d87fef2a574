"""TEST INFRASTRUCTURE ONLY: collectives for several ranks sharing ONE GPU (the test box has one): a gloo group whose
device tensors are staged through host memory.  The product's collectives are torch.distributed on device tensors under
backend "nccl" (= RCCL, mindrec_amd.wide_deep._DirectComm); this class lets the sharded engine's protocol run with the
real HIP kernels on one card."""
import torch
import torch.distributed as dist


class StagedGlooComm:
    def __init__(self, group=None):
        self.group = group

    def all_to_all(self, out, inp, out_splits=None, in_splits=None):
        if out.dtype in (torch.bfloat16, torch.float16):          # gloo has no 16-bit floats: ship the bytes (row splits unchanged)
            o8 = torch.empty(out.shape[:-1] + (out.shape[-1] * 2,), dtype=torch.uint8)
            dist.all_to_all_single(o8, inp.cpu().contiguous().view(torch.uint8), out_splits, in_splits, group=self.group)
            out.copy_(o8.view(out.dtype))
            return
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=self.group)
        out.copy_(o)

    def all_to_all_lists(self, outs, ins):
        """Per-peer tensor lists (rows of one width and dtype, any row counts) through one staged uneven all-to-all."""
        ref = next(t for t in list(ins) + list(outs))
        tail = tuple(ref.shape[1:])
        send = torch.cat([t.reshape((-1,) + tail) for t in ins]) if ins else ref[:0]
        recv = torch.empty((sum(t.shape[0] for t in outs),) + tail, dtype=ref.dtype, device=ref.device)
        self.all_to_all(recv, send, [t.shape[0] for t in outs], [t.shape[0] for t in ins])
        o = 0
        for t in outs:
            t.copy_(recv[o:o + t.shape[0]])
            o += t.shape[0]

    def all_reduce(self, t, async_op=False):
        c = t.cpu()
        dist.all_reduce(c, group=self.group)
        t.copy_(c)
        return None
