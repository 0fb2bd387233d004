"""Per-kernel parity: each HIP entry point (called through the C ABI via codon_amd.ops) against
the same op stated with torch CPU fp32 (the oracle's building blocks).  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.util import rel_rmse, rmse


def _dev():
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(size=shape) * scale).astype(np.float32))


def _act(t, dev):
    """(B,C,H,W) CPU tensor -> device activation buffer in the layout the kernels use for its dtype (16-bit:
    channel-blocked, codon_amd/csrc/c8.h)."""
    from codon_amd import ops
    return ops.from_nchw(t.to(dev))


def _new(B, C, H, W, dtype, dev):
    from codon_amd import ops
    buf = ops.new_act(B, C, H, W, dtype, dev)
    buf.fill_(float("nan"))
    return buf


def _nchw(buf):
    from codon_amd import ops
    return ops.to_nchw(buf).float().cpu()


CONV_CASES = [  # (k, cin, cout)
    (5, 128, 128), (5, 64, 64), (3, 64, 64), (3, 128, 64), (1, 128, 64), (3, 64, 128), (1, 64, 128),
]
SHAPES = [(2, 19, 45), (1, 8, 32), (1, 33, 70), (1, 1, 1), (1, 5, 3), (3, 16, 64)]


@pytest.mark.parametrize("k,cin,cout", CONV_CASES)
@pytest.mark.parametrize("shape", SHAPES)
def test_conv2d_vs_torch(k, cin, cout, shape):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x = _rand((B, cin, H, W), 1)
    w = _rand((cout, cin, k, k), 2, scale=(2.0 / (k * k * cout)) ** 0.5)
    ref = F.conv2d(x, w, None, 1, k // 2)
    xd, wd = x.to(dev), w.to(dev)
    wp = ops.packed_weight(wd)
    y = torch.full((B, cout, H, W), float("nan"), device=dev)
    ops.conv2d(Slice(xd), wp, Slice(y), k)
    assert rel_rmse(y.cpu(), ref) < 2e-6
    # fused ReLU
    ops.conv2d(Slice(xd), wp, Slice(y), k, relu=True)
    assert rel_rmse(y.cpu(), F.relu(ref)) < 2e-6
    # residual add
    r = _rand((B, cout, H, W), 3)
    ops.conv2d(Slice(xd), wp, Slice(y), k, residual=Slice(r.to(dev)))
    assert rel_rmse(y.cpu(), ref + r) < 2e-6


def test_conv2d_channel_slices():
    """Reads channels [64,128) of a 128-ch buffer, writes channels [64,128) of another: the
    cat-free layout of CODON_x4.py:79-80."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = 2, 21, 37
    xb = _rand((B, 128, H, W), 5)
    w = _rand((64, 64, 5, 5), 6, 0.05)
    yb = torch.full((B, 128, H, W), 7.0)
    ref = F.relu(F.conv2d(xb[:, 64:], w, None, 1, 2))
    xd, yd = xb.to(dev), yb.to(dev)
    ops.conv2d(Slice(xd, 64, 64), ops.packed_weight(w.to(dev)), Slice(yd, 64, 64), 5, relu=True)
    got = yd.cpu()
    assert torch.equal(got[:, :64], yb[:, :64])  # untouched half
    assert rel_rmse(got[:, 64:], ref) < 2e-6


def test_conv2d_dgrad_pack():
    """PACK_DGRAD weights turn the forward kernel into dL/dx."""
    from codon_amd import _lib as L
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    for (k, cin, cout) in [(5, 64, 64), (3, 128, 64), (1, 128, 64), (5, 128, 128)]:
        B, H, W = 1, 13, 40
        x = _rand((B, cin, H, W), 1).requires_grad_(True)
        w = _rand((cout, cin, k, k), 2, 0.05)
        gy = _rand((B, cout, H, W), 3)
        F.conv2d(x, w, None, 1, k // 2).backward(gy)
        wp = ops.packed_weight(w.to(dev), L.PACK_DGRAD)
        gx = torch.empty((B, cin, H, W), device=dev)
        ops.conv2d(Slice(gy.to(dev)), wp, Slice(gx), k)
        assert rel_rmse(gx.cpu(), x.grad) < 2e-6, (k, cin, cout)


@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 16, 64), (1, 1, 1), (2, 7, 5)])
def test_stem_head(shape):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x = _rand((B, 1, H, W), 1)
    w = _rand((64, 1, 3, 3), 2, 0.3)
    ref = F.relu(F.conv2d(x, w, None, 1, 1))
    y = torch.empty((B, 64, H, W), device=dev)
    ops.stem(x.to(dev), w.to(dev), Slice(y))
    assert rel_rmse(y.cpu(), ref) < 1e-6
    f = _rand((B, 64, H, W), 3)
    wo = _rand((1, 64, 3, 3), 4, 0.1)
    refo = F.conv2d(f, wo, None, 1, 1) + x
    o = torch.empty((B, 1, H, W), device=dev)
    ops.head(Slice(f.to(dev)), wo.to(dev), x.to(dev), o)
    assert rel_rmse(o.cpu(), refo) < 1e-6


@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 48, 64), (1, 1, 1), (3, 50, 41), (1, 64, 96)])
def test_cac_gate_kernels(shape):
    from codon_amd import ops
    from codon_amd.ops import Slice
    from oracle import codon_oracle as orc
    dev = _dev()
    B, H, W = shape
    pre2 = _rand((B, 128, H, W), 1)           # [pre | pre_c]
    in2 = _rand((B, 128, H, W), 2)
    w1, b1 = _rand((8, 128), 3, 0.1), _rand((8,), 4, 0.1)
    w2, b2 = _rand((64, 8), 5, 0.3), _rand((64,), 6, 0.1)
    ws = _rand((1, 2, 5, 5), 7, 0.2)
    pre, pre_c = pre2[:, :64], pre2[:, 64:]
    Fcat = torch.cat((pre_c, pre), 1)
    ch_ref = orc.cac_channel(Fcat, w1, b1, w2, b2)
    sp_ref = orc.cac_spatial(Fcat, ws)
    g = ch_ref[:, :, None, None] * sp_ref
    out_ref, outc_ref = pre * g + in2[:, :64], pre_c * g + in2[:, 64:]

    p2, i2 = pre2.to(dev), in2.to(dev)
    nt = ops.cac_stats_tiles(H, W)
    pooled = torch.empty((B, 2, H, W), device=dev)
    partials = torch.empty((B, nt, 128, 2), device=dev)
    ops.cac_stats(Slice(p2, 64, 64), Slice(p2, 0, 64), pooled, partials)
    assert rel_rmse(pooled[:, 0].cpu(), Fcat.max(1)[0]) == 0.0
    assert rel_rmse(pooled[:, 1].cpu(), Fcat.mean(1)) < 1e-6
    ch = torch.empty((B, 64), device=dev)
    pools = torch.empty((B, 2, 128), device=dev)
    ops.cac_gate(B, H, W, partials, w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), ch, pools)
    assert rel_rmse(pools[:, 0].cpu(), Fcat.mean((2, 3))) < 1e-5 or rmse(pools[:, 0].cpu(), Fcat.mean((2, 3))) < 1e-6
    assert torch.equal(pools[:, 1].cpu(), Fcat.amax((2, 3)))
    assert rmse(ch.cpu(), ch_ref) < 1e-6
    sp = torch.empty((B, 1, H, W), device=dev)
    ops.cac_spatial(pooled, ws.to(dev), sp)
    assert rmse(sp.cpu(), sp_ref) < 1e-6
    oc = torch.empty((B, 128, H, W), device=dev)
    ops.cac_apply(Slice(p2, 0, 64), Slice(p2, 64, 64), ch, sp, Slice(i2, 0, 64), Slice(i2, 64, 64),
                  Slice(oc, 0, 64), Slice(oc, 64, 64))
    assert rel_rmse(oc[:, :64].cpu(), out_ref) < 1e-6
    assert rel_rmse(oc[:, 64:].cpu(), outc_ref) < 1e-6


def test_error_conventions():
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    x = torch.zeros((1, 48, 4, 4), device=dev)
    y = torch.zeros((1, 64, 4, 4), device=dev)
    with pytest.raises(RuntimeError, match="no f32 kernel"):
        ops.conv2d(Slice(x), torch.zeros(48 * 64 * 9, device=dev), Slice(y), 3)
    with pytest.raises(RuntimeError, match="HIP device"):
        ops.conv2d(Slice(x.cpu()), torch.zeros(8), Slice(y.cpu()), 3)


WGRAD_CASES = [(5, 128, 128), (5, 64, 64), (3, 64, 64), (3, 128, 64), (1, 128, 64)]


@pytest.mark.parametrize("k,cin,cout", WGRAD_CASES)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 4, 32), (1, 1, 1), (3, 9, 70), (2, 21, 44), (1, 10, 72)])
def test_conv2d_wgrad_vs_autograd(k, cin, cout, shape):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x = _rand((B, cin, H, W), 1)
    w = _rand((cout, cin, k, k), 2, 0.05).requires_grad_(True)
    gy = _rand((B, cout, H, W), 3)
    F.conv2d(x, w, None, 1, k // 2).backward(gy)
    dw = torch.full((cout, cin, k, k), float("nan"), device=dev)
    ops.conv2d_wgrad(Slice(x.to(dev)), Slice(gy.to(dev)), dw, k)
    assert rel_rmse(dw.cpu(), w.grad) < 3e-6
    # accumulate (shared weights): dw += same again
    ops.conv2d_wgrad(Slice(x.to(dev)), Slice(gy.to(dev)), dw, k, accumulate=True)
    assert rel_rmse(dw.cpu(), 2 * w.grad) < 3e-6
    # deterministic: fixed-order reduction
    dw2 = torch.empty_like(dw)
    ops.conv2d_wgrad(Slice(x.to(dev)), Slice(gy.to(dev)), dw2, k)
    dw3 = torch.empty_like(dw)
    ops.conv2d_wgrad(Slice(x.to(dev)), Slice(gy.to(dev)), dw3, k)
    assert torch.equal(dw2, dw3)


def test_conv2d_wgrad_slices_and_bands():
    """Channel slices of wider buffers + an image tall enough to be split into several bands."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = 1, 130, 40
    xb = _rand((B, 128, H, W), 1)
    gb = _rand((B, 128, H, W), 2)
    w = torch.zeros((64, 64, 5, 5), requires_grad=True)
    F.conv2d(xb[:, 64:], w, None, 1, 2).backward(gb[:, :64])
    dw = torch.empty((64, 64, 5, 5), device=dev)
    ops.conv2d_wgrad(Slice(_act(xb, dev), 64, 64), Slice(_act(gb, dev), 0, 64), dw, 5)
    assert rel_rmse(dw.cpu(), w.grad) < 3e-6


def test_conv2d_relu_mask_and_accumulate():
    """dgrad through a ReLU: gx = (x_saved > 0) ? conv(gy, w') : 0, accumulated into a fan-in buffer."""
    from codon_amd import _lib as L
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = 2, 11, 37
    xs = torch.relu(_rand((B, 64, H, W), 1))
    w = _rand((64, 64, 3, 3), 2, 0.1)
    gy = _rand((B, 64, H, W), 3)
    xv = xs.clone().requires_grad_(True)
    F.conv2d(xv, w, None, 1, 1).backward(gy)
    ref = xv.grad * (xs > 0)
    base = _rand((B, 64, H, W), 4)
    out = base.to(dev)
    ops.conv2d(Slice(gy.to(dev)), ops.packed_weight(w.to(dev), L.PACK_DGRAD), Slice(out), 3,
               relu_mask=Slice(xs.to(dev)), accumulate=True)
    assert rel_rmse(out.cpu(), base + ref) < 2e-6


# ---- bf16 path (BASELINE configs[2], [4]): bf16 activations / weights, fp32 accumulate --------------

BF16_TOL = 1e-2   # rel-RMSE of one bf16 conv vs the fp32 op on bf16-rounded operands is ~3e-3 (output rounding)


@pytest.mark.parametrize("k,cin,cout", CONV_CASES)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 8, 32), (1, 1, 1), (1, 33, 70)])
def test_conv2d_bf16_vs_torch(k, cin, cout, shape):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x = _rand((B, cin, H, W), 1).bfloat16()
    w = _rand((cout, cin, k, k), 2, scale=(2.0 / (k * k * cout)) ** 0.5)
    ref = F.conv2d(x.float(), w.bfloat16().float(), None, 1, k // 2)     # same operand rounding, fp32 accumulate
    wp = ops.packed_weight(w.to(dev), dtype=torch.bfloat16)
    y = _new(B, cout, H, W, torch.bfloat16, dev)
    xd = _act(x, dev)
    ops.conv2d(Slice(xd), wp, Slice(y), k)
    assert rel_rmse(_nchw(y), ref) < 3e-3            # only the output rounding to bf16 differs
    r = _rand((B, cout, H, W), 3).bfloat16()
    ops.conv2d(Slice(xd), wp, Slice(y), k, relu=True)
    assert rel_rmse(_nchw(y), F.relu(ref)) < 3e-3
    ops.conv2d(Slice(xd), wp, Slice(y), k, residual=Slice(_act(r, dev)))
    assert rel_rmse(_nchw(y), ref + r.float()) < 3e-3
    # backward epilogues: ReLU mask of the tensor the gradient flows into, then fan-in accumulate
    ops.conv2d(Slice(xd), wp, Slice(y), k, relu_mask=Slice(_act(r, dev)))
    masked = torch.where(r.float() > 0, ref, torch.zeros_like(ref))
    assert rel_rmse(_nchw(y), masked) < 3e-3
    ops.conv2d(Slice(xd), wp, Slice(y), k, accumulate=True)
    assert rel_rmse(_nchw(y), masked.bfloat16().float() + ref) < 4e-3


def test_conv2d_bf16_channel_slices():
    """Slices of wider channel-blocked buffers (how the torch.cat calls of CODON_x4.py:79,80,119 disappear)."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = 2, 21, 37
    xb = _rand((B, 128, H, W), 1).bfloat16()
    w = _rand((64, 64, 3, 3), 2, scale=0.06)
    wp = ops.packed_weight(w.to(dev), dtype=torch.bfloat16)
    yb = _new(B, 128, H, W, torch.bfloat16, dev)
    yb.zero_()
    ops.conv2d(Slice(_act(xb, dev), 64, 64), wp, Slice(yb, 64, 64), 3, relu=True)
    ref = F.relu(F.conv2d(xb[:, 64:].float(), w.bfloat16().float(), None, 1, 1))
    out = _nchw(yb)
    assert rel_rmse(out[:, 64:], ref) < 3e-3 and float(out[:, :64].abs().max()) == 0.0


def test_bf16_elementwise_kernels():
    from codon_amd import ops
    from codon_amd.ops import Slice
    from oracle import codon_oracle as orc
    dev = _dev()
    for (B, H, W) in [(2, 16, 24), (1, 19, 45), (1, 1, 1)]:
        x = _rand((B, 1, H, W), 1)
        w = _rand((64, 1, 3, 3), 2, 0.3)
        y = _new(B, 64, H, W, torch.bfloat16, dev)
        ops.stem(x.to(dev), w.to(dev), Slice(y))
        assert rel_rmse(_nchw(y), F.relu(F.conv2d(x, w, None, 1, 1))) < 3e-3
        f = _rand((B, 64, H, W), 3).bfloat16()
        wo = _rand((1, 64, 3, 3), 4, 0.1)
        o = torch.empty((B, 1, H, W), device=dev)
        ops.head(Slice(_act(f, dev)), wo.to(dev), x.to(dev), o)
        assert rel_rmse(o.cpu(), F.conv2d(f.float(), wo, None, 1, 1) + x) < 1e-5
        pre2 = _rand((B, 128, H, W), 5).bfloat16()
        in2 = _rand((B, 128, H, W), 6).bfloat16()
        w1, b1, w2, b2 = _rand((8, 128), 7, 0.1), _rand((8,), 8, 0.1), _rand((64, 8), 9, 0.3), _rand((64,), 10, 0.1)
        ws = _rand((1, 2, 5, 5), 11, 0.2)
        pre, pre_c = pre2[:, :64].float(), pre2[:, 64:].float()
        Fcat = torch.cat((pre_c, pre), 1)
        ch_ref, sp_ref = orc.cac_channel(Fcat, w1, b1, w2, b2), orc.cac_spatial(Fcat, ws)
        g = ch_ref[:, :, None, None] * sp_ref
        p2, i2 = _act(pre2, dev), _act(in2, dev)
        nt = ops.cac_stats_tiles(H, W)
        pooled = torch.empty((B, 2, H, W), device=dev); partials = torch.empty((B, nt, 128, 2), device=dev)
        ch = torch.empty((B, 64), device=dev); sp = torch.empty((B, 1, H, W), device=dev)
        ops.cac_stats(Slice(p2, 64, 64), Slice(p2, 0, 64), pooled, partials)
        assert torch.equal(pooled[:, 0].cpu(), Fcat.max(1)[0])
        ops.cac_gate(B, H, W, partials, w1.to(dev), b1.to(dev), w2.to(dev), b2.to(dev), ch, None)
        ops.cac_spatial(pooled, ws.to(dev), sp)
        assert rmse(ch.cpu(), ch_ref) < 1e-6 and rmse(sp.cpu(), sp_ref) < 1e-6
        oc = _new(B, 128, H, W, torch.bfloat16, dev)
        ops.cac_apply(Slice(p2, 0, 64), Slice(p2, 64, 64), ch, sp, Slice(i2, 0, 64), Slice(i2, 64, 64),
                      Slice(oc, 0, 64), Slice(oc, 64, 64))
        ocn = _nchw(oc)
        assert rel_rmse(ocn[:, :64], pre * g + in2[:, :64].float()) < 3e-3
        assert rel_rmse(ocn[:, 64:], pre_c * g + in2[:, 64:].float()) < 3e-3


@pytest.mark.parametrize("k,cin,cout", WGRAD_CASES)
# (1, 65, 65): W % 32 == 1 and H % 4 == 1 -- the 3x3 kernel's border-free fast path must not take the tile whose staged
# columns tx0+33, tx0+34 lie past the row end (they wrapped into the next row / past the end of x)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 4, 32), (1, 1, 1), (3, 9, 70), (1, 130, 40), (3, 13, 48), (2, 40, 8), (1, 65, 65)])
def test_conv2d_wgrad_bf16_vs_autograd(k, cin, cout, shape):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x = _rand((B, cin, H, W), 1).bfloat16()
    gy = _rand((B, cout, H, W), 3).bfloat16()
    w = torch.zeros((cout, cin, k, k), requires_grad=True)
    F.conv2d(x.float(), w, None, 1, k // 2).backward(gy.float())      # same bf16-rounded operands, fp32 math
    dw = torch.full((cout, cin, k, k), float("nan"), device=dev)
    ops.conv2d_wgrad(Slice(_act(x, dev)), Slice(_act(gy, dev)), dw, k)
    assert dw.dtype == torch.float32
    assert rel_rmse(dw.cpu(), w.grad) < 1e-5        # fp32 accumulate: only summation order differs
    ops.conv2d_wgrad(Slice(_act(x, dev)), Slice(_act(gy, dev)), dw, k, accumulate=True)
    assert rel_rmse(dw.cpu(), 2 * w.grad) < 1e-5


@pytest.mark.parametrize("W", [37, 40])
def test_conv2d_wgrad_bf16_slices(W):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H = 2, 21
    xb = _rand((B, 128, H, W), 1).bfloat16()
    gb = _rand((B, 128, H, W), 2).bfloat16()
    w = torch.zeros((64, 64, 5, 5), requires_grad=True)
    F.conv2d(xb[:, 64:].float(), w, None, 1, 2).backward(gb[:, :64].float())
    dw = torch.empty((64, 64, 5, 5), device=dev)
    ops.conv2d_wgrad(Slice(_act(xb, dev), 64, 64), Slice(_act(gb, dev), 0, 64), dw, 5)
    assert rel_rmse(dw.cpu(), w.grad) < 1e-5


@pytest.mark.parametrize("k,cin,cout", [(5, 128, 128), (3, 64, 64), (1, 128, 64)])
def test_conv2d_fp16_vs_torch(k, cin, cout):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = 2, 19, 46
    x = _rand((B, cin, H, W), 1).half()
    w = _rand((cout, cin, k, k), 2, scale=(2.0 / (k * k * cout)) ** 0.5)
    ref = F.conv2d(x.float(), w.half().float(), None, 1, k // 2)
    wp = ops.packed_weight(w.to(dev), dtype=torch.float16)
    y = _new(B, cout, H, W, torch.float16, dev)
    ops.conv2d(Slice(_act(x, dev)), wp, Slice(y), k, relu=True)
    assert rel_rmse(_nchw(y), F.relu(ref)) < 5e-4


def test_conv2d_plane_larger_than_32bit_slice():
    """One 3000x3000 image: a 128-channel slice would be 4.6 GB, past a single 32-bit buffer descriptor -- the
    kernels re-base their descriptors per channel chunk / cout tile.  Checked on three row windows against torch."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    H = W = 3000
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand((1, 128, H, W), generator=g, device=dev) - 0.5
    w = _rand((64, 128, 3, 3), 6, scale=0.03)
    y = torch.empty((1, 64, H, W), device=dev)
    ops.conv2d(Slice(x), ops.packed_weight(w.to(dev)), Slice(y), 3, relu=True)
    for r0 in (0, 1480, H - 24):
        lo, hi = max(r0 - 1, 0), min(r0 + 25, H)
        ref = F.relu(F.conv2d(x[:, :, lo:hi].cpu(), w, None, 1, 1))[:, :, r0 - lo:r0 - lo + 24]
        assert rel_rmse(y[:, :, r0:r0 + 24].cpu(), ref) < 2e-6
    del x, y
    torch.cuda.empty_cache()
