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
    """Owns the 44 parameter gradients of a CODONNet as views of ONE flat buffer and all-reduces it.

    Two routes fill the buffer, with the same values:
      * the ordinary one -- `loss.backward()`, `torch.autograd.grad(...)`, `loss.backward(inputs=[...])`: the backward
        RETURNS the gradients autograd asked for and autograd accumulates them into `.grad` (= the views).  Nothing is
        written that the caller did not ask for: `torch.autograd.grad(loss, [x])` leaves `self.flat` untouched.
      * the direct one, OPT-IN PER BACKWARD CALL -- `gs.backward(loss)` (or `with gs.direct_backward(): loss.backward()`):
        the backward's own kernels ADD the parameter gradients into the views (codon_amd.autograd._grad_sink) and return
        None for them -- no per-tensor AccumulateGrad add, no ATen kernel in the step.  Inside that context EVERY backward
        through the model adds into `.grad`, whatever `inputs=` it was given: use it around the training step's
        `loss.backward()` only.  direct=False (or CODON_GRAD_DIRECT=0) makes `gs.backward` the ordinary route too (A/B).
    Anything that breaks the aliasing (optimizer.zero_grad() with set_to_none=True, a foreign .grad) silently falls back to the
    ordinary route for that step."""

    def __init__(self, model, process_group: Optional[dist.ProcessGroup] = None, direct: bool = True):
        import weakref
        self.group = process_group
        import os
        self.direct = bool(direct) and os.environ.get("CODON_GRAD_DIRECT", "1") != "0"      # 0: A/B of the ordinary route
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
        self.view_ptrs = [p.grad.data_ptr() for p in self.params]
        self._unused = [p for p in model.parameters() if all(p is not q for q in self.params)]
        self._model = model
        self._armed = 0                            # > 0 only inside direct_backward(): the backward may ADD into the views
        self._host = None                          # pinned staging copy of the flat buffer (gloo only)
        model.__dict__["_grad_sink"] = weakref.ref(self)      # not a submodule, not pickled (model.__getstate__ drops it)

    def direct_backward(self):
        """Context: backward passes through the model started inside it add the parameter gradients straight into the flat
        buffer (see the class docstring).  Thread-confined like the backward itself; re-entrant."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            self._armed += 1
            try:
                yield self
            finally:
                self._armed -= 1
        return ctx()

    def backward(self, loss, **kw):
        """`loss.backward(**kw)` on the direct route (the training step's call)."""
        with self.direct_backward():
            loss.backward(**kw)

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
        if self.flat.is_cuda and dist.get_backend(self.group) == "gloo" and not async_op:
            # gloo (the one-GPU rehearsals and the fallback when RCCL cannot form a communicator): through a pinned host copy
            # and gloo's CPU algorithm.  ProcessGroupGloo's own device-tensor path takes SECONDS per call, sporadically, when
            # the ranks share one card (measured round 6: the 4-rank rehearsal 6 s or 180-235 s; 25 ms per call this way)
            if self._host is None:
                self._host = torch.empty(self.flat.shape, dtype=self.flat.dtype, pin_memory=True)
            self._host.copy_(self.flat, non_blocking=False)
            dist.all_reduce(self._host, op=dist.ReduceOp.SUM, group=self.group)
            self.flat.copy_(self._host, non_blocking=False)
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)


class FlatAdam:
    """torch.optim.Adam (amsgrad=False, maximize=False) over a GradSync's parameters as ONE launch per step
    (codon_adam_step): the gradients are read from the flat all-reduce buffer, the two moments live in flat buffers of the
    same layout, the fp32 parameters are updated in place (and their Tensor._version bumped: the packed-weight cache and
    GraphedCODON.stale() see the new values).  Same update as torch.optim.Adam to fp32 rounding (tests/test_gpu_reduce.py)."""

    def __init__(self, gs: GradSync, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0):
        import ctypes as C
        from . import _lib as L
        if gs.flat.dtype != torch.float32 or not gs.flat.is_cuda or any(not p.is_contiguous() for p in gs.params):
            raise NotImplementedError("FlatAdam: fp32 parameters on the GPU")
        if len(gs.params) > L.ADAM_MAX:
            raise NotImplementedError(f"FlatAdam: at most {L.ADAM_MAX} tensors")
        self.gs, self.lr, self.betas, self.eps, self.weight_decay = gs, float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(gs.flat), torch.zeros_like(gs.flat)
        self.t = 0
        self._C, self._L = C, L

    def step(self):
        C, L, gs = self._C, self._L, self.gs
        gs._install_views()                        # a dropped / foreign .grad is adopted into the flat buffer first
        d = L.AdamDesc()
        d.n = len(gs.params)
        for i, p in enumerate(gs.params):
            d.param[i], d.count[i] = p.data_ptr(), p.numel()
        self.t += 1
        dev = gs.flat.device
        with torch.cuda.device(dev):
            L.check(L.load().codon_adam_step(C.byref(d), C.c_void_p(gs.flat.data_ptr()), C.c_void_p(self.exp_avg.data_ptr()),
                                             C.c_void_p(self.exp_avg_sq.data_ptr()), self.lr, self.betas[0], self.betas[1],
                                             self.eps, self.weight_decay, self.t,
                                             C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "adam_step")
        for p in gs.params:                        # written through raw pointers: make the new values visible to version keys
            torch.autograd.graph.increment_version(p)


def shard_batch(n_images: int, rank: int, world: int):
    """Contiguous, balanced image range of `rank` (units = images; no image is split)."""
    base, rem = divmod(n_images, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def grad_equality_selfcheck(device, size=(24, 20), tol: float = 2e-5, group: Optional[dist.ProcessGroup] = None):
    """First-contact check of the data-parallel path on whatever backend the process group runs (RCCL over xGMI on the
    8-GPU node; gloo in rehearsals): SURVEY.md 8(e)'s correctness test on the product kernels, small enough to run before
    every multi-rank benchmark.  Ranks build DIFFERENT parameters, rank 0's are broadcast, every rank runs the HIP forward
    + backward on ITS image of a world-sized batch (per-image-mean L1 loss), one all-reduce of the flat gradient; then
    each rank computes the gradient of the whole batch by itself and compares, in fp32 and in bf16.  Returns
    {"grad_equal": bool over all ranks, "worst_rel": {dtype: max over tensors and ranks}, "first_forward_equal": bool}.
    group: the process group the gradients travel on (None = the default group)."""
    import numpy as np
    from .model import CODONNet
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    H, W = size
    rng = np.random.default_rng(4242)
    x = torch.from_numpy(rng.uniform(0, 1, size=(world, 1, H, W)).astype(np.float32)).to(device)
    y = torch.from_numpy(rng.uniform(0, 1, size=(world, 1, H, W)).astype(np.float32)).to(device)
    t = torch.from_numpy(rng.uniform(0, 1, size=(world, 1, H, W)).astype(np.float32)).to(device)
    state = torch.random.get_rng_state()
    torch.manual_seed(1000 + rank)
    m = CODONNet().to(device).train()
    torch.random.set_rng_state(state)
    with torch.no_grad():
        m(x[:1], y[:1])                                   # packs this rank's OWN weights: the broadcast must drop them
    gs = GradSync(m, process_group=group)
    gs.broadcast_parameters(0)
    worst, ok = {}, True
    with torch.no_grad():
        o = m(x[:1], y[:1]).float()
    digest = torch.stack([o.double().sum(), o.double().abs().sum(), o.double().pow(2).sum()])
    lo_, hi_ = digest.clone(), digest.clone()
    if world > 1:
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN, group=group)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX, group=group)
    first_equal = bool(torch.equal(lo_, hi_))
    for dt, tag in ((None, "f32"), (torch.bfloat16, "bf16")):
        m.set_compute_dtype(dt)
        gs.zero_grad()
        gs.backward((m(x[rank:rank + 1], y[rank:rank + 1]) - t[rank:rank + 1]).abs().mean())
        gs.all_reduce_grads()
        avg = gs.flat.clone()
        gs.zero_grad()
        # The single-process side takes the SUM of the per-image-mean losses, so every image gets the upstream gradient
        # sign / (H W) it gets on its own rank -- bit for bit the same 16-bit activation gradients on both sides for ANY world
        # size (the mean over the whole batch, sign / (world H W), rounds differently in bf16 unless 1 / world is a power of
        # two: ADVICE r5) -- and the fp32 parameter gradients are scaled by 1 / world afterwards.  What is left between the
        # two sides is the order of the fp32 sums over images and where the 1 / world factor is applied.
        gs.backward((m(x, y) - t).abs().mean(dim=(1, 2, 3)).sum())
        gs.flat.mul_(1.0 / world)
        w, off = 0.0, 0
        for p in gs.params:
            n = p.numel()
            a, b = avg[off:off + n].double(), gs.flat[off:off + n].double()
            w = max(w, float((a - b).norm() / (b.norm() + 1e-30)))
            off += n
        wt = torch.tensor([w], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(wt, op=dist.ReduceOp.MAX, group=group)
        worst[tag] = float(wt.item())
        ok = ok and worst[tag] <= tol and bool(torch.isfinite(avg).all())
    m.check_packed()
    return {"grad_equal": bool(ok and first_equal), "first_forward_equal": first_equal, "worst_rel": worst, "tol": tol,
            "tol_bf16": tol,
            "ranks": world, "what": "N-rank averaged HIP gradient vs the same rank's single-process HIP gradient on the "
                                    "concatenated batch, worst tensor over all ranks (SURVEY.md 8e)"}
