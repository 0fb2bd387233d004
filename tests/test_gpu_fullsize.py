"""Parity at the FULL BASELINE.json workloads (configs[1..4] = C2..C5), where 32-bit offsets, grid limits and
plane re-basing actually matter (5 GB per 128-channel tensor at C2, 10 GB at C5).

The CPU oracle takes seconds per 480x640 image and ~2 min per 1920x2560 image, so each test compares the LAST image
of the full batch (and the first, at C2) against the oracle run on that single image, and the whole batch against
single-image HIP runs bit for bit (images are independent units): together they pin every image of the
batch to the oracle.

  C2  x4  fwd  b32  480x640   fp32   /root/reference/CODON_X4/CODON_x4.py:66-132
  C3  x4  fwd+bwd b32/GPU 480x640 bf16 (per-GPU shape of the 8-GPU config)
  C4  x8  fwd  b16  960x1280  fp32   /root/reference/CODON_X8/CODON_x8.py (same net as x4)
  C5  x16 fwd  b8   1920x2560 bf16   /root/reference/CODON_X16/CODON_x16.py:136-202

Tolerances: fp32 RMSE <= 1e-4 absolute (north_star); bf16 rel-RMSE <= 3e-2 vs the fp32 oracle (the reference's own
bf16 CPU run sits at 1.8e-2); gradients: whole vector <= 1e-4 (fp32) / <= 2e-2 vs the oracle and linearity <= 2e-3 (bf16)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import codon_oracle as orc
from tests.util import rel_rmse, rmse, target_for


def _threads():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(min(n, 16))       # the GPU box grants a 16-core share; oneDNN oversubscribed runs 3x slower


def _inputs(B, H, W, seed):
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.random((B, 1, H, W), dtype=np.float32))
    y = torch.from_numpy((g.integers(0, 256, size=(B, 1, H, W)) / 255.0).astype(np.float32))
    return x, y


def _model(variant, sd):
    from codon_amd import CODONNet, CODONNet16
    m = (CODONNet16 if variant == "x16" else CODONNet)()
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def _free():
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


def test_c2_x4_b32_480x640_fp32_first_and_last_image_vs_oracle():
    _threads()
    B, H, W = 32, 480, 640
    sd = orc.he_state("x4", seed=40)
    x, y = _inputs(B, H, W, 41)
    m = _model("x4", sd)
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
        assert torch.isfinite(o).all()
        for i in (0, B - 1):
            single = m(x[i:i + 1].cuda(), y[i:i + 1].cuda())
            assert torch.equal(o[i:i + 1], single), f"image {i} of the batch differs from its single-image run"
            ref = orc.forward(sd, x[i:i + 1], y[i:i + 1])
            e = rmse(single.cpu(), ref)
            assert e <= 1e-4, (i, e)
            assert rel_rmse(single.cpu(), ref) <= 2e-5
        # every other image: bit-identical to a chunked run (4 x 8 images) -- no offset aliasing anywhere in the batch
        chunks = torch.cat([m(x[j:j + 8].cuda(), y[j:j + 8].cuda()) for j in range(0, B, 8)])
        assert torch.equal(o, chunks)
    del o, chunks, m
    _free()


def test_c4_x8_b16_960x1280_fp32_last_image_vs_oracle():
    _threads()
    B, H, W = 16, 960, 1280
    sd = orc.he_state("x8", seed=42)
    x, y = _inputs(B, H, W, 43)
    m = _model("x8", sd)
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
        assert torch.isfinite(o).all()
        i = B - 1
        single = m(x[i:i + 1].cuda(), y[i:i + 1].cuda())
        assert torch.equal(o[i:i + 1], single)
        assert torch.equal(o[:1], m(x[:1].cuda(), y[:1].cuda()))
        halves = torch.cat([m(x[:8].cuda(), y[:8].cuda()), m(x[8:].cuda(), y[8:].cuda())])
        assert torch.equal(o, halves)
        ref = orc.forward(sd, x[i:i + 1], y[i:i + 1])          # ~25 s on 16 host threads
    e = rmse(single.cpu(), ref)
    assert e <= 1e-4, e
    assert rel_rmse(single.cpu(), ref) <= 2e-5
    del o, halves, m
    _free()


def test_c5_x16_b8_1920x2560_bf16_last_image_vs_oracle():
    _threads()
    B, H, W = 8, 1920, 2560
    sd = orc.he_state("x16", seed=44)
    x, y = _inputs(B, H, W, 45)
    m = _model("x16", sd).set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
        assert o.dtype == torch.float32 and torch.isfinite(o).all()
        i = B - 1
        single = m(x[i:i + 1].cuda(), y[i:i + 1].cuda())
        assert torch.equal(o[i:i + 1], single)
        assert torch.equal(o[:1], m(x[:1].cuda(), y[:1].cuda()))
        halves = torch.cat([m(x[:4].cuda(), y[:4].cuda()), m(x[4:].cuda(), y[4:].cuda())])
        assert torch.equal(o, halves)
        single = single.cpu()
        del o, halves, m
        _free()
        ref = orc.forward(sd, x[i:i + 1], y[i:i + 1])          # fp32 oracle, ~2 min on 16 host threads
    e = rel_rmse(single, ref)
    assert e <= 3e-2, e


def _grad_vector(m):
    from codon_amd.autograd import used_parameters
    return torch.cat([p.grad.detach().float().flatten() for _, p in used_parameters(m)])


def test_c3_x4_b32_480x640_bf16_training_step_linearity():
    """One full-size bf16 forward+backward (117 GB of saved activations): finite gradients, and -- gradients being
    linear in the per-image upstream gradient -- grad(batch of 32) == grad(first 16) + grad(last 16) when each run is
    fed its slice of the same upstream gradient (fp32 accumulation order is the only difference)."""
    B, H, W = 32, 480, 640
    sd = orc.he_state("x4", seed=46)
    x, y = _inputs(B, H, W, 47)
    g = torch.from_numpy(np.sign(np.random.default_rng(48).standard_normal((B, 1, H, W))).astype(np.float32)) / (B * H * W)
    m = _model("x4", sd).set_compute_dtype(torch.bfloat16)
    m.train()
    xc, yc, gc = x.cuda(), y.cuda(), g.cuda()

    def run(lo, hi):
        m.zero_grad(set_to_none=True)
        out = m(xc[lo:hi].contiguous(), yc[lo:hi].contiguous())
        out.backward(gc[lo:hi].contiguous())
        v = _grad_vector(m)
        del out
        _free()
        return v

    full = run(0, B)
    assert torch.isfinite(full).all() and float(full.abs().max()) > 0
    parts = run(0, 16) + run(16, B)
    e = float((full.double() - parts.double()).norm() / parts.double().norm())
    assert e <= 2e-3, e
    del m
    _free()


def test_c3_shape_fp32_gradient_of_one_image_vs_oracle_autograd():
    """fp32 backward of one full 480x640 image (configs[2]'s image size) against the CPU oracle's autograd
    (~10 s on the GPU box's 16 host threads)."""
    _threads()
    B, H, W = 1, 480, 640
    sd = orc.he_state("x4", seed=49)
    x, y = _inputs(B, H, W, 50)
    tgt = target_for(x)
    loss_ref, gref, out_ref = orc.grads(sd, x, y, tgt)
    m = _model("x4", sd)
    out = m(x.cuda(), y.cuda())
    assert rmse(out.detach().cpu(), out_ref) <= 1e-4
    g_up = (torch.sign(out_ref - tgt) / out_ref.numel()).cuda()
    out.backward(g_up)
    num = den = 0.0
    for k, p in m.named_parameters():
        if k in gref:
            num += float((p.grad.cpu().double() - gref[k].double()).pow(2).sum())
            den += float(gref[k].double().pow(2).sum())
    assert (num / den) ** 0.5 <= 1e-4, (num / den) ** 0.5
    del m, out
    _free()
    # ... and the bf16 backward (the dtype of the metric's fwd+bwd half) at the same size against the same oracle
    # gradient, same upstream gradient -- an oracle comparison at full size, not only a property.  Measured on MI355X:
    # whole vector 7.6e-3, worst conv tensor (>= 36 864 elements) 1.4e-2 (conv5.weight); at this size the sums run over
    # 307 200 pixels and the 16-bit rounding noise averages out further than in the 2x24x20 fixtures (where the
    # reference's own bf16 autograd sits at 1-3e-2 on such tensors, tests/golden/bf16grad_*.npz)
    mb = _model("x4", sd).set_compute_dtype(torch.bfloat16)
    mb.train()
    outb = mb(x.cuda(), y.cuda())
    assert rel_rmse(outb.detach().float().cpu(), out_ref) <= 3e-2
    outb.backward(g_up)
    num = den = 0.0
    worst = ("", 0.0)
    for k, p in mb.named_parameters():
        if k in gref:
            d2 = float((p.grad.cpu().double() - gref[k].double()).pow(2).sum())
            r2 = float(gref[k].double().pow(2).sum())
            num += d2
            den += r2
            if p.numel() >= 36864 and (d2 / r2) ** 0.5 > worst[1]:
                worst = (k, (d2 / r2) ** 0.5)
    print(f"[1x480x640 bf16 backward vs oracle fp32 autograd] whole vector {(num / den) ** 0.5:.3e}, worst large tensor {worst}")
    assert (num / den) ** 0.5 <= 2e-2, (num / den) ** 0.5
    assert worst[1] <= 4e-2, worst
    del mb, outb
    _free()
