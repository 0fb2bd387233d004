"""Data-parallel training support: one process per GPU, images sharded across ranks, ONE all-reduce of
a flat gradient buffer per step (RCCL over xGMI on MI355X; gloo on CPU in tests).

The reference's only multi-GPU construct is single-process torch.nn.DataParallel
(/root/reference/CODON_X16/test.py:52): scatter inputs, re-broadcast every parameter to every replica
on every forward, gather outputs on GPU 0.  Nothing of that is reproduced.  Forward needs no
collective at all (no op mixes samples).  For training, the 44 used parameter tensors (1 865 506
values, 7.46 MB fp32 -- latency-bound on xGMI) get their .grad laid out as views into
one contiguous buffer, so the collective is a single call with no packing copies.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .autograd import used_parameters


class GradSync:
    def __init__(self, model, process_group: Optional[dist.ProcessGroup] = None):
        self.group = process_group
        self.named = used_parameters(model)
        self.params = [p for _, p in self.named]
        dev, dt = self.params[0].device, self.params[0].dtype
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        off = 0
        for p in self.params:                      # .grad of every used parameter = a view of the flat buffer
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n
        self._unused = [p for p in model.parameters() if all(p is not q for q in self.params)]
        self._model = model

    def _install_views(self):
        off = 0
        for p in self.params:
            n = p.numel()
            v = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                # dropped by optimizer.zero_grad(set_to_none=True): the slice still holds the previous step's averaged
                # gradient -- clear it, or a parameter that receives no gradient this step would inherit the old one
                v.zero_()
                p.grad = v
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)                    # a gradient produced outside the buffer: adopt it, then alias
                p.grad = v
            off += n

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    def broadcast_parameters(self, src: int = 0):
        """Once, before training: every rank starts from rank `src`'s parameters (all 49/44 tensors)."""
        if not dist.is_initialized() or self.world_size == 1:
            return
        with torch.no_grad():
            for p in list(self.params) + self._unused:
                dist.broadcast(p, src=src, group=self.group)
                # a collective writes the tensor without bumping Tensor._version (measured: gloo, torch 2.10); bump it
                # so that everything keyed on (data_ptr, _version) -- the packed-weight cache, GraphedCODON.stale() --
                # sees the new values
                torch.autograd.graph.increment_version(p)
        if hasattr(self._model, "invalidate_packed"):
            self._model.invalidate_packed()        # packed MFMA weight images of the old values must not survive

    def zero_grad(self):
        """Use this (or optimizer.zero_grad(set_to_none=False)) -- NOT optimizer.zero_grad() with its default
        set_to_none=True, which drops the .grad views into the flat buffer; all_reduce_grads() re-installs the
        views if that happened, at the price of one copy per tensor."""
        self.flat.zero_()
        self._install_views()

    def all_reduce_grads(self, async_op: bool = False):
        """Average the flat gradient over ranks.  With a per-image-mean loss on equal shards this equals
        the single-process gradient on the concatenated batch."""
        self._install_views()                      # no-op when every .grad still aliases the flat buffer
        if not dist.is_initialized() or self.world_size == 1:
            return None
        self.flat.mul_(1.0 / self.world_size)      # pre-scale: the SUM then is the mean, one pass
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)


def shard_batch(n_images: int, rank: int, world: int):
    """Contiguous, balanced image range of `rank` (units = images; no image is split)."""
    base, rem = divmod(n_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
