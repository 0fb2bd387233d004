"""conv5x5(128->128)+ReLU with the 1x1 (128->64) [+ residual] chained from the accumulators
(codon_conv_chain1x1_fwd) against the same two ops in torch CPU fp32 and against the two separate HIP launches."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.test_gpu_kernels import _dev, _rand
from tests.util import rel_rmse

SHAPES = [(2, 19, 45), (1, 8, 32), (1, 33, 70), (1, 1, 1), (1, 5, 3), (3, 16, 64)]


def _case(shape):
    B, H, W = shape
    x = _rand((B, 128, H, W), 11)
    w5 = _rand((128, 128, 5, 5), 12, scale=(2.0 / (25 * 128)) ** 0.5)
    w1 = _rand((64, 128, 1, 1), 13, scale=(1.0 / 128) ** 0.5)
    r = _rand((B, 64, H, W), 14)
    mid = F.relu(F.conv2d(x, w5, None, 1, 2))
    return x, w5, w1, r, mid, F.conv2d(mid, w1)


@pytest.mark.parametrize("shape", SHAPES)
def test_chain1x1_f32_vs_torch(shape):
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x, w5, w1, r, mid_ref, out_ref = _case(shape)
    xd = x.to(dev)
    wp = ops.packed_weight(w5.to(dev))
    wc = ops.packed_weight(w1.to(dev), L.PACK_CHAIN1X1)
    # outputs land in channels [64,128) of a wider buffer (the Fcat layout); the rest must stay untouched
    out = torch.full((B, 128, H, W), float("nan"), device=dev)
    ops.conv_chain1x1(Slice(xd), wp, wc, Slice(out, 64, 64))
    assert torch.isnan(out[:, :64]).all()
    assert rel_rmse(out[:, 64:].cpu(), out_ref) < 2e-6
    # + residual, and the intermediate materialised (what training saves)
    mid = torch.full((B, 128, H, W), float("nan"), device=dev)
    o2 = torch.full((B, 64, H, W), float("nan"), device=dev)
    ops.conv_chain1x1(Slice(xd), wp, wc, Slice(o2), mid=Slice(mid), residual=Slice(r.to(dev)))
    assert rel_rmse(o2.cpu(), out_ref + r) < 2e-6
    assert rel_rmse(mid.cpu(), mid_ref) < 2e-6
    # the two separate launches: the intermediate is bit-identical, the 1x1 differs by summation order only
    m2 = torch.empty_like(mid)
    ops.conv2d(Slice(xd), wp, Slice(m2), 5, relu=True)
    assert torch.equal(m2, mid)
    o3 = torch.empty_like(o2)
    ops.conv2d(Slice(m2), ops.packed_weight(w1.to(dev)), Slice(o3), 1, residual=Slice(r.to(dev)))
    assert rel_rmse(o2.cpu(), o3.cpu()) < 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
def test_chain1x1_16bit(shape, dtype):
    """16-bit tensors: same bar as the separate 16-bit launches (operands rounded to 16 bits, fp32 accumulate), and
    the materialised intermediate is bit-identical to what the unfused conv stores."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x, w5, w1, r, _, _ = _case(shape)
    q = lambda t: t.to(dtype).float()
    mid_ref = q(F.relu(F.conv2d(q(x), q(w5), None, 1, 2)))
    out_ref = F.conv2d(mid_ref, q(w1)) + q(r)
    xd, rd = ops.from_nchw(x.to(dev), dtype), ops.from_nchw(r.to(dev), dtype)      # channel-blocked 16-bit buffers
    wp = ops.packed_weight(w5.to(dev), L.PACK_FWD, dtype)
    wc = ops.packed_weight(w1.to(dev), L.PACK_CHAIN1X1, dtype)
    nan = lambda c: ops.new_act(B, c, H, W, dtype, dev).fill_(float("nan"))
    nchw = lambda buf: ops.to_nchw(buf).float().cpu()
    mid, out = nan(128), nan(128)
    ops.conv_chain1x1(Slice(xd), wp, wc, Slice(out, 0, 64), mid=Slice(mid), residual=Slice(rd))
    assert torch.isnan(nchw(out)[:, 64:]).all()
    tol = 6e-3 if dtype == torch.bfloat16 else 8e-4
    assert rel_rmse(nchw(out)[:, :64], out_ref) < tol
    assert rel_rmse(nchw(mid), mid_ref) < tol
    m2 = torch.empty_like(mid)
    ops.conv2d(Slice(xd), wp, Slice(m2), 5, relu=True)
    assert torch.equal(m2, mid)
    o3 = nan(64)
    ops.conv2d(Slice(m2), ops.packed_weight(w1.to(dev), L.PACK_FWD, dtype), Slice(o3), 1, residual=Slice(rd))
    assert rel_rmse(nchw(out)[:, :64], nchw(o3)) < tol
    # without the intermediate
    o4 = nan(64)
    ops.conv_chain1x1(Slice(xd), wp, wc, Slice(o4), residual=Slice(rd))
    assert torch.equal(ops.to_nchw(o4), ops.to_nchw(out)[:, :64])


@pytest.mark.parametrize("shape", SHAPES)
def test_chain1x1_f16x3(shape):
    """opt-in split-precision evaluation of fp32 tensors: both stages as 3 f16 MFMAs per product."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x, w5, w1, r, mid_ref, out_ref = _case(shape)
    xd = x.to(dev)
    wp = ops.packed_weight(w5.to(dev), L.PACK_FWD_F16X3)
    wc = ops.packed_weight(w1.to(dev), L.PACK_CHAIN1X1_F16X3)
    mid = torch.full((B, 128, H, W), float("nan"), device=dev)
    out = torch.full((B, 128, H, W), float("nan"), device=dev)
    ops.conv_chain1x1(Slice(xd), wp, wc, Slice(out, 64, 64), mid=Slice(mid), residual=Slice(r.to(dev)), f16x3=True)
    assert torch.isnan(out[:, :64]).all()
    assert rel_rmse(out[:, 64:].cpu(), out_ref + r) < 4e-6
    assert rel_rmse(mid.cpu(), mid_ref) < 4e-6
    m2 = torch.empty_like(mid)
    ops.conv2d(Slice(xd), wp, Slice(m2), 5, relu=True, f16x3=True)
    assert torch.equal(m2, mid)
    o4 = torch.empty((B, 64, H, W), device=dev)
    ops.conv_chain1x1(Slice(xd), wp, wc, Slice(o4), f16x3=True)
    assert rel_rmse(o4.cpu(), out_ref) < 4e-6


def test_chain1x1_rejects_bad_arguments():
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    x = torch.zeros((1, 64, 4, 4), device=dev)
    w = torch.zeros(64 * 64 * 25, device=dev)
    with pytest.raises(RuntimeError):
        ops.conv_chain1x1(Slice(x), w, w, Slice(torch.zeros((1, 64, 4, 4), device=dev)))     # cin must be 128
    with pytest.raises(RuntimeError):
        ops.packed_weight(torch.zeros((64, 64, 1, 1), device=dev), L.PACK_CHAIN1X1)           # (64,128,1,1) only


@pytest.mark.parametrize("k,cin", [(5, 64), (3, 64), (3, 128)])
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 33, 70), (1, 1, 1), (1, 5, 3), (3, 16, 64)])
def test_gated_conv_equals_apply_then_conv(shape, k, cin):
    """codon_conv2d_gated_fwd (the CAC gate-apply formed in the conv's staging) is bit-identical to
    codon_cac_apply_fwd followed by codon_conv2d_fwd, and equals torch within the fp32 kernel bar."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    coff = 0 if cin == 128 else 64                                   # the colour-stream slice when cin = 64
    pre2, in2 = _rand((B, 128, H, W), 31), _rand((B, 128, H, W), 32)
    ch = torch.sigmoid(_rand((B, 64), 33))
    sp = torch.sigmoid(_rand((B, 1, H, W), 34))
    w = _rand((64, cin, k, k), 35, scale=(2.0 / (k * k * 64)) ** 0.5)
    pd, idv, chd, spd = pre2.to(dev), in2.to(dev), ch.to(dev), sp.to(dev)
    wp = ops.packed_weight(w.to(dev))
    y = torch.full((B, 64, H, W), float("nan"), device=dev)
    ops.conv2d_gated(Slice(pd, coff, cin), Slice(idv, coff, cin), chd, spd, wp, Slice(y), k, relu=True)
    oc = torch.empty((B, 128, H, W), device=dev)
    ops.cac_apply(Slice(pd, 0, 64), Slice(pd, 64, 64), chd, spd, Slice(idv, 0, 64), Slice(idv, 64, 64),
                  Slice(oc, 0, 64), Slice(oc, 64, 64))
    y2 = torch.empty_like(y)
    ops.conv2d(Slice(oc, coff, cin), wp, Slice(y2), k, relu=True)
    assert torch.equal(y, y2)
    # emit (codon_conv2d_gated_emit_fwd): the conv also writes the gated input it staged, every own pixel once, into a
    # slice of a wider buffer -- bit for bit cac_apply's output
    y3 = torch.full((B, 64, H, W), float("nan"), device=dev)
    xg = torch.full((B, 64 + cin + 8, H, W), float("nan"), device=dev)
    ops.conv2d_gated(Slice(pd, coff, cin), Slice(idv, coff, cin), chd, spd, wp, Slice(y3), k, relu=True,
                     emit=Slice(xg, 64, cin))
    assert torch.equal(y, y3)
    assert torch.equal(xg[:, 64:64 + cin], oc[:, coff:coff + cin])
    assert torch.isnan(xg[:, :64]).all() and torch.isnan(xg[:, 64 + cin:]).all()
    g = ch[:, :, None, None] * sp                                     # (B,64,H,W)
    xin = pre2 * torch.cat([g, g], 1) + in2
    ref = F.relu(F.conv2d(xin[:, coff:coff + cin], w, None, 1, k // 2))
    assert rel_rmse(y.cpu(), ref) < 2e-6
