"""Multi-rank logic on CPU with gloo (world_size 2): gradient averaging over a flat buffer and image
sharding.  The compute stand-in is the CPU oracle (the HIP kernels need a GPU); what is under test is
codon_amd.dist: N-rank averaged gradients == 1-rank gradients on the concatenated batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from codon_amd import CODONNet
    from codon_amd.autograd import used_parameters
    from codon_amd.dist import GradSync, shard_batch
    from oracle import codon_oracle as orc
    from tests.util import target_for
    torch.manual_seed(100 + rank)                       # ranks start from DIFFERENT parameters
    m = CODONNet()
    gs = GradSync(m)
    assert gs.numel == 1865506 and len(gs.params) == 44
    v0 = [p._version for p in m.parameters()]
    gs.broadcast_parameters(src=0)
    # the collective itself does not bump Tensor._version; broadcast_parameters must (GraphedCODON.stale()
    # and the packed-weight cache key on it), for the unused attention_*5 tensors as well
    assert all(p._version > v for p, v in zip(m.parameters(), v0))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    # global batch of 4 images, sharded by image
    B, H, W = 4, 12, 10
    x, y = orc.kat_inputs(B, H, W)
    tgt = target_for(x)
    lo, hi = shard_batch(B, rank, world)
    _, g, _ = orc.grads(sd, x[lo:hi], y[lo:hi], tgt[lo:hi])      # per-shard mean loss
    gs.zero_grad()
    for n, p in gs.named:
        p.grad.add_(g[n])                                # autograd accumulates in place into the flat views
    gs.all_reduce_grads()
    if rank == 0:
        _, gfull, _ = orc.grads(sd, x, y, tgt)           # single process, concatenated batch
        errs = {n: float((p.grad - gfull[n]).norm() / (gfull[n].norm() + 1e-30)) for n, p in gs.named}
        q.put((errs, float(gs.flat.norm()), [tuple(shard_batch(7, r, 3)) for r in range(3)]))
    else:
        q.put(float(sum(float(v.double().sum()) for v in sd.values())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errs, flat_norm, shards = next(r for r in res if isinstance(r, tuple))
    assert flat_norm > 0
    assert shards == [(0, 3), (3, 5), (5, 7)]
    bad = {k: v for k, v in errs.items() if v > 2e-5}
    assert not bad, bad


def test_flat_views_and_single_process_noop():
    from codon_amd import CODONNet16
    from codon_amd.dist import GradSync
    m = CODONNet16()
    gs = GradSync(m)
    assert gs.numel == 1865506
    assert m.conv3.weight.grad.data_ptr() != 0 and m.conv3.weight.grad._base is gs.flat
    m.conv3.weight.grad.fill_(2.0)
    assert float(gs.flat.sum()) == 2.0 * m.conv3.weight.numel()
    assert gs.all_reduce_grads() is None               # no process group: nothing to do
    gs.zero_grad()
    assert float(gs.flat.abs().sum()) == 0.0


def test_grad_views_survive_set_to_none():
    """optimizer.zero_grad() defaults to set_to_none=True, which drops the .grad views into the flat buffer;
    the next all_reduce_grads() (or gs.zero_grad()) must adopt whatever gradients exist and re-install the views, so the
    collective never reduces stale zeros."""
    from codon_amd import CODONNet16
    from codon_amd.dist import GradSync
    m = CODONNet16()
    gs = GradSync(m)
    opt = torch.optim.SGD(gs.params, lr=0.1)
    opt.zero_grad()                                     # set_to_none=True: every .grad is None now
    assert m.conv3.weight.grad is None
    m.conv3.weight.grad = torch.full_like(m.conv3.weight, 3.0)       # a gradient produced OUTSIDE the buffer
    m.confuse.weight.grad = torch.full_like(m.confuse.weight, 1.0)
    gs.all_reduce_grads()                               # single process: no collective, but the views are repaired
    assert m.conv3.weight.grad._base is gs.flat and m.input.weight.grad._base is gs.flat
    assert float(gs.flat.sum()) == 3.0 * m.conv3.weight.numel() + m.confuse.weight.numel()
    gs.zero_grad()
    assert float(gs.flat.abs().sum()) == 0.0 and m.conv3.weight.grad._base is gs.flat


def test_broadcast_invalidates_packed_weights():
    """GradSync.broadcast_parameters must drop packed conv weights (single process: the no-op path leaves them)."""
    from codon_amd import CODONNet16
    from codon_amd.dist import GradSync
    m = CODONNet16()
    gs = GradSync(m)
    m._pack_cache["probe"] = ("tag", torch.zeros(1))
    gs.broadcast_parameters(0)                          # no process group -> returns early, cache untouched
    assert "probe" in m._pack_cache
    m.invalidate_packed()
    assert not m._pack_cache


def test_dropped_grad_slice_is_cleared_not_inherited():
    """after optimizer.zero_grad() (set_to_none) a parameter that gets NO gradient in the next step must not
    inherit the previous step's averaged gradient from its slice of the flat buffer."""
    from codon_amd import CODONNet16
    from codon_amd.dist import GradSync
    m = CODONNet16()
    gs = GradSync(m)
    gs.flat.fill_(5.0)                                  # "last step's" gradients
    opt = torch.optim.SGD(gs.params, lr=0.1)
    opt.zero_grad()                                     # every .grad is None; the flat buffer still holds the 5s
    m.conv3.weight.grad = torch.full_like(m.conv3.weight, 2.0)       # only conv3 gets a gradient this step
    gs.all_reduce_grads()
    assert float(m.conv3.weight.grad.sum()) == 2.0 * m.conv3.weight.numel()
    assert float(gs.flat.sum()) == 2.0 * m.conv3.weight.numel()      # nothing of the 5s survives
    assert float(m.confuse.weight.grad.abs().sum()) == 0.0 and m.confuse.weight.grad._base is gs.flat
