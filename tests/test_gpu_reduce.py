"""Round 5: every fixed-order reduction of a backward pass in ONE launch (codon_reduce_multi), and parameter gradients ADDED
straight into codon_amd.dist.GradSync's flat buffer by the backward's own kernels.  What autograd of
/root/reference/CODON_X4/CODON_x4.py:66-132 produces is unchanged: the deferred form must equal the immediate form (one
small reduce behind each producer) BIT FOR BIT, and that one is pinned to the oracle / golden gradients elsewhere
(tests/test_gpu_backward.py)."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import codon_oracle as orc
from tests.util import rel_rmse, target_for


def _model(sd, dtype=None):
    from codon_amd import CODONNet
    m = CODONNet()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    if dtype is not None:
        m.set_compute_dtype(dtype)
    return m


def test_reduce_multi_items_against_float64():
    """wgrad items (1, 3 and 7 uses of a weight: more than CODON_REDUCE_MAX_USES chains a second item), rows items serial /
    16 chunks / 64 chunks with row counts that are NOT multiples of the chunk count, the 3x3 tap flip, accumulate on and off."""
    from codon_amd import ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(3)
    for accumulate in (False, True):
        red = ops.DeferredReduce()
        outs, want = {}, {}

        def base(key, shape):
            outs[key] = torch.randn(shape, generator=g).to(dev)
            return outs[key].double().cpu() if accumulate else torch.zeros(shape, dtype=torch.float64)

        for key, (co, ci, k, nsplit, nuse) in {"a": (64, 64, 3, 9, 1), "b": (128, 128, 5, 4, 3), "c": (64, 128, 1, 33, 7)}.items():
            ref = base(key, (co, ci, k, k))
            for _ in range(nuse):
                ws = torch.randn((nsplit, k * k, co, ci), generator=g)
                red.add_wgrad(key, ws.to(dev).reshape(-1), co, ci, k * k)
                ref = ref + ws.double().sum(0).permute(1, 2, 0).reshape(co, ci, k, k)
            want[key] = ref
        for key, (n, nparts, stride, off, nchunk, flip) in {"r1": (1024, 32, 1608, 0, 1, False), "r2": (8, 32, 1608, 1024, 1, False),
                                                             "r3": (50, 300, 50, 0, 64, False), "r4": (50, 63, 50, 0, 1, False),
                                                             "r5": (576, 37, 576, 0, 16, True), "r6": (576, 5, 576, 0, 16, False),
                                                             "r7": (50, 9600, 50, 0, 64, False)}.items():
            ref = base(key, (n,))
            part = torch.randn((nparts * stride + 2048,), generator=g)
            red.add_rows(key, part.to(dev), off, n, nparts, stride, nchunk=nchunk, flip9=flip)
            rows = part[: nparts * stride].reshape(nparts, stride)[:, off:off + n].double().sum(0)
            if flip:
                rows = rows.reshape(-1, 9).flip(1).reshape(-1)
            want[key] = ref + rows
        red.run(outs, accumulate=accumulate)
        torch.cuda.synchronize()
        for key in want:
            assert rel_rmse(outs[key].cpu().double(), want[key]) < 3e-6, (key, accumulate)


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_deferred_reduce_equals_the_immediate_form_bit_for_bit(dtype):
    from codon_amd import autograd
    sd = orc.he_state("x4", seed=23)
    rng = np.random.default_rng(5)
    B, H, W = 3, 37, 70
    x = torch.from_numpy(rng.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32)).cuda()
    up = torch.from_numpy(rng.standard_normal(size=(B, 1, H, W)).astype(np.float32)).cuda() / (B * H * W)
    grads = []
    old = autograd.DEFER_REDUCE
    try:
        for defer in (False, True):
            autograd.DEFER_REDUCE = defer
            m = _model(sd, dtype)
            m(x, y).backward(up)
            torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    finally:
        autograd.DEFER_REDUCE = old
    assert len(grads[0]) == 44 == len(grads[1])
    bad = [k for k in grads[0] if not torch.equal(grads[0][k], grads[1][k])]
    assert not bad, bad


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
@pytest.mark.parametrize("defer", [True, False])
def test_gradients_added_into_the_gradsync_buffer_equal_returned_gradients(dtype, defer):
    """gs.backward(loss) / `with gs.direct_backward()`: the backward kernels add into the .grad views themselves.  Same values
    as the ordinary route bit for bit (0 + g == g), a second backward accumulates (g + g), and a dropped view
    (zero_grad(set_to_none=True)) falls back to the ordinary route for that step.  Outside the context the
    ordinary route runs: a plain loss.backward() gives the same bits, and torch.autograd.grad / backward(inputs=[x]) write
    nothing into the flat buffer."""
    from codon_amd import autograd
    from codon_amd.dist import GradSync
    sd = orc.he_state("x4", seed=29)
    x, y = orc.kat_inputs(2, 24, 40)
    x, y = x.cuda(), y.cuda()
    tgt = target_for(x.cpu()).cuda()
    old = autograd.DEFER_REDUCE
    try:
        autograd.DEFER_REDUCE = defer
        m0 = _model(sd, dtype)
        (m0(x, y) - tgt).abs().mean().backward()
        ref = {k: p.grad.clone() for k, p in m0.named_parameters() if p.grad is not None}
        m = _model(sd, dtype)
        gs = GradSync(m)
        assert autograd._grad_sink(m) is None              # opt-in per backward call
        with gs.direct_backward():
            assert autograd._grad_sink(m) is not None
        assert autograd._grad_sink(m) is None
        # the ordinary route through a GradSync-owned model: only what was asked for is written
        gs.zero_grad()
        xr = x.clone().requires_grad_(True)
        gx, = torch.autograd.grad((m(xr, y) - tgt).abs().mean(), [xr])
        assert gx is not None and bool(torch.isfinite(gx).all()) and float(gs.flat.abs().max()) == 0.0
        (m(xr, y) - tgt).abs().mean().backward(inputs=[xr])
        assert float(gs.flat.abs().max()) == 0.0 and torch.equal(xr.grad, gx)
        with gs.direct_backward():                          # ... and the same request inside the context does what it says
            pass
        (m(x, y) - tgt).abs().mean().backward()            # plain backward: autograd accumulates into the views
        bad = [n for n, p in gs.named if not torch.equal(p.grad, ref[n])]
        assert not bad, bad
        assert all(p.grad._base is gs.flat for p in gs.params)
        calls = []
        hooks = [p.register_post_accumulate_grad_hook(lambda p_: calls.append(1)) for p in gs.params]
        gs.zero_grad()
        gs.backward((m(x, y) - tgt).abs().mean())
        torch.cuda.synchronize()
        assert len(calls) == 44                # post-accumulate hooks still fire (AccumulateGrad sees "no gradient": no add)
        names = [n for n, _ in gs.named]
        assert len(names) == 44
        bad = [n for n, p in gs.named if not torch.equal(p.grad, ref[n])]
        assert not bad, bad
        assert all(p.grad._base is gs.flat for p in gs.params)
        # a second backward accumulates: ((g + r_1) + r_2) + ... over the uses of a shared weight -- g + g up to fp32
        # re-association (exactly g + g for the single-use tensors)
        gs.backward((m(x, y) - tgt).abs().mean())
        bad = [n for n, p in gs.named if rel_rmse(p.grad.cpu(), (ref[n] + ref[n]).cpu()) > 1e-6]
        assert not bad, bad
        assert torch.equal(m.conv7.weight.grad, ref["conv7.weight"] + ref["conv7.weight"])
        # the unused tensors stay without gradient
        assert m.attention_c5.mlp[1].weight.grad is None
        # fallback: one view dropped -> the whole step goes through autograd's accumulation, then the views are re-adopted
        for h in hooks:
            h.remove()
        gs.zero_grad()
        m.conv3.weight.grad = None
        with gs.direct_backward():
            assert autograd._grad_sink(m) is None
        gs.backward((m(x, y) - tgt).abs().mean())
        gs.all_reduce_grads()
        bad = [n for n, p in gs.named if not torch.equal(p.grad, ref[n])]
        assert not bad, bad
        with gs.direct_backward():
            assert autograd._grad_sink(m) is not None
        # direct=False never takes the sink
        m2 = _model(sd, dtype)
        gs2 = GradSync(m2, direct=False)
        with gs2.direct_backward():
            assert autograd._grad_sink(m2) is None
        gs2.backward((m2(x, y) - tgt).abs().mean())
        assert all(torch.equal(p.grad, ref[n]) for n, p in gs2.named)
    finally:
        autograd.DEFER_REDUCE = old


def test_cac_spatial_weight_gradient_with_a_block_count_that_is_not_a_multiple_of_64():
    """110 blocks (1 x 320 x 352): per = 2, 55 chunks hold rows.  The immediate form's second stage used to walk all 64
    chunk slots and read past the partial buffer; both forms against float64."""
    from codon_amd import _lib as L, ops
    dev = torch.device("cuda:0")
    B, H, W = 1, 320, 352
    g = torch.Generator().manual_seed(11)
    g_z = torch.randn((B, 1, H, W), generator=g)
    pooled = torch.randn((B, 2, H, W), generator=g)
    w = torch.randn((1, 2, 5, 5), generator=g)
    lib = L.load()
    nsb = lib.codon_cac_bwd_spatial_blocks(B, H, W)
    assert nsb == 110
    pz = torch.nn.functional.pad(pooled.double(), (2, 2, 2, 2))
    want = torch.stack([(g_z[0, 0].double() * pz[0, c, dy:dy + H, dx:dx + W]).sum() for c in range(2) for dy in range(5)
                        for dx in range(5)])
    gzd, pd_, wd = g_z.to(dev), pooled.to(dev), w.to(dev)
    for defer in (False, True):
        # poison what follows the partial rows: an out-of-range read shows
        part = torch.full((nsb * 50 + 4096,), float("nan"), device=dev)
        g_pooled = torch.empty((B, 2, H, W), device=dev)
        dws = torch.empty((50,), device=dev)
        with torch.cuda.device(dev):
            L.check(lib.codon_cac_bwd_spatial(B, H, W, C.c_void_p(gzd.data_ptr()), C.c_void_p(pd_.data_ptr()),
                                              C.c_void_p(wd.data_ptr()), C.c_void_p(g_pooled.data_ptr()),
                                              C.c_void_p(part.data_ptr()), None if defer else C.c_void_p(dws.data_ptr()),
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "cac_bwd_spatial")
        if defer:
            red = ops.DeferredReduce()
            red.add_rows("w", part, 0, 50, nsb, 50, nchunk=64)
            red.run({"w": dws}, accumulate=False)
        torch.cuda.synchronize()
        assert torch.isfinite(dws).all(), defer
        assert rel_rmse(dws.cpu().double(), want) < 2e-6, defer


@pytest.mark.parametrize("wd", [0.0, 1e-2])
def test_flat_adam_equals_torch_optim_adam(wd):
    """codon_amd.dist.FlatAdam (codon_adam_step: one launch over all 44 tensors, gradients read from GradSync's flat buffer)
    against torch.optim.Adam on the same parameters and gradients, five steps: parameters and both moments to fp32 rounding.
    The packed-weight cache must see the update (Tensor._version is bumped)."""
    from codon_amd.dist import FlatAdam, GradSync
    sd = orc.he_state("x4", seed=31)
    ma, mb = _model(sd), _model(sd)
    gsa, gsb = GradSync(ma), GradSync(mb)
    opt_a = FlatAdam(gsa, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    opt_b = torch.optim.Adam(gsb.params, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    g = torch.Generator(device="cpu").manual_seed(5)
    v0 = [p._version for p in gsa.params]
    for step in range(5):
        grad = (torch.randn(gsa.numel, generator=g) * (0.5 + step)).cuda()
        gsa.flat.copy_(grad)
        gsb.flat.copy_(grad)
        opt_a.step()
        opt_b.step()
    torch.cuda.synchronize()
    assert all(p._version > v for p, v in zip(gsa.params, v0))
    worst = 0.0
    for (n, pa), pb in zip(gsa.named, gsb.params):
        worst = max(worst, rel_rmse(pa.detach().cpu(), pb.detach().cpu()))
        assert rel_rmse(pa.detach().cpu(), pb.detach().cpu()) <= 1e-6, n
        # the update itself (what Adam added), not only the parameter it is small against
        d_a, d_b = (pa.detach() - sd[n].cuda()).cpu(), (pb.detach() - sd[n].cuda()).cpu()
        assert rel_rmse(d_a, d_b) <= 1e-4, n
    off = 0
    for p, pb in zip(gsa.params, gsb.params):
        st = opt_b.state[pb]
        k = p.numel()
        assert rel_rmse(opt_a.exp_avg[off:off + k].cpu(), st["exp_avg"].flatten().cpu()) <= 1e-6
        assert rel_rmse(opt_a.exp_avg_sq[off:off + k].cpu(), st["exp_avg_sq"].flatten().cpu()) <= 1e-6
        off += k
    # a forward after the step runs on re-packed weights (no stale trip), and equals the torch-Adam twin's to fp32 noise
    x, y = orc.kat_inputs(1, 24, 20)
    with torch.no_grad():
        oa, ob = ma.eval()(x.cuda(), y.cuda()), mb.eval()(x.cuda(), y.cuda())
    ma.check_packed(); mb.check_packed()
    assert rel_rmse(oa.cpu(), ob.cpu()) <= 1e-5
