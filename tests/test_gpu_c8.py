"""Channel-blocked 16-bit kernels (codon_amd/csrc/{ew_c8,conv_c8,conv_wgrad_c8}.hip) against the fp32 NCHW kernels of
the same entry points on the SAME values (inputs rounded to bf16 / fp16 first, so the two paths see identical operands
and differ only by the 16-bit rounding of what they store).  The fp32 kernels themselves are pinned to torch / the oracle
in test_gpu_kernels.py and test_gpu_backward.py.  GPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tests.util import rel_rmse, rmse

DT = [torch.bfloat16, torch.float16]
SHAPES = [(2, 19, 45), (1, 1, 1), (1, 33, 70), (2, 64, 40), (1, 5, 3)]


def _dev():
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch.device("cuda:0")


def _rand(shape, seed, scale=1.0):
    g = np.random.default_rng(seed)
    return torch.from_numpy((g.standard_normal(size=shape) * scale).astype(np.float32))


def _tol(dtype):
    return 4e-3 if dtype == torch.bfloat16 else 6e-4


def test_layout_round_trip_and_addressing():
    """ops.from_nchw / to_nchw state the layout of csrc/c8.h: element (b,c,h,w) at (((b*C/8 + c/8)*H + h)*W + w)*8 + c%8."""
    from codon_amd import ops
    B, C, H, W = 2, 24, 3, 5
    t = torch.arange(B * C * H * W, dtype=torch.float32).reshape(B, C, H, W) % 251
    buf = ops.from_nchw(t, torch.bfloat16)
    assert tuple(buf.shape) == (B, C // 8, H, W, 8) and buf.is_contiguous()
    flat = buf.flatten()
    for (b, c, h, w) in [(0, 0, 0, 0), (1, 23, 2, 4), (0, 9, 1, 3), (1, 8, 0, 0)]:
        assert float(flat[(((b * (C // 8) + c // 8) * H + h) * W + w) * 8 + c % 8]) == float(t[b, c, h, w])
    assert torch.equal(ops.to_nchw(buf).float(), t)
    s = ops.Slice(buf, 8, 16)
    assert s.ctotal == 24 and torch.equal(s.view().float(), t[:, 8:24])
    with pytest.raises(AssertionError):
        ops.Slice(buf, 4, 8)                       # slices start on a plane boundary
    with pytest.raises(AssertionError):
        ops.Slice(t.bfloat16())                    # an NCHW 16-bit tensor is not an activation buffer


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", SHAPES)
def test_stem_and_masked_flipped_stencil(shape, dtype):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x = _rand((B, 1, H, W), 1).to(dev)
    w = _rand((64, 1, 3, 3), 2, 0.3).to(dev)
    mask = _rand((B, 64, H, W), 3).to(dtype).float().to(dev)
    for relu, flip, use_mask in [(True, False, False), (False, True, True), (False, False, True)]:
        ref = torch.empty((B, 128, H, W), device=dev)
        ops.stencil_1to64(x, w, Slice(ref, 64, 64), relu=relu, flip=flip, mask=Slice(mask) if use_mask else None)
        out = ops.new_act(B, 128, H, W, dtype, dev).fill_(float("nan"))
        ops.stencil_1to64(x, w, Slice(out, 64, 64), relu=relu, flip=flip,
                          mask=Slice(ops.from_nchw(mask, dtype)) if use_mask else None)
        o = ops.to_nchw(out).float()
        assert torch.isnan(o[:, :64]).all()
        assert torch.equal(o[:, 64:], ref[:, 64:].to(dtype).float())       # same fp32 arithmetic, one rounding


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", SHAPES + [(1, 40, 130)])
def test_head_and_conv1ch_wgrad(shape, dtype):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    f = _rand((B, 128, H, W), 3).to(dtype).float().to(dev)
    wo = _rand((1, 64, 3, 3), 4, 0.1).to(dev)
    res = _rand((B, 1, H, W), 5).to(dev)
    ref = torch.empty((B, 1, H, W), device=dev)
    ops.head(Slice(f, 64, 64), wo, res, ref)
    out = torch.full((B, 1, H, W), float("nan"), device=dev)
    ops.head(Slice(ops.from_nchw(f, dtype), 64, 64), wo, res, out)
    assert rel_rmse(out.cpu(), ref.cpu()) < 2e-6               # fp32 math on identical operands: summation order only
    tref = F.conv2d(f[:, 64:].cpu(), wo.cpu(), None, 1, 1) + res.cpu()
    assert rel_rmse(out.cpu(), tref) < 2e-6
    s = _rand((B, 1, H, W), 6).to(dev)
    for flip in (False, True):
        d0, d1 = torch.empty(576, device=dev), torch.full((576,), float("nan"), device=dev)
        ops.conv1ch_wgrad(Slice(f, 0, 64), s, d0, flip=flip)
        ops.conv1ch_wgrad(Slice(ops.from_nchw(f, dtype), 0, 64), s, d1, flip=flip)
        assert rel_rmse(d1.cpu(), d0.cpu()) < 1e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", SHAPES)
def test_cac_forward_passes(shape, dtype):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    pre2 = _rand((B, 128, H, W), 5).to(dtype).float().to(dev)
    in2 = _rand((B, 128, H, W), 6).to(dtype).float().to(dev)
    p16, i16 = ops.from_nchw(pre2, dtype), ops.from_nchw(in2, dtype)
    chs = torch.rand((B, 64), device=dev)
    nt = ops.cac_stats_tiles(H, W)
    mk = lambda: (torch.empty((B, 2, H, W), device=dev), torch.empty((B, nt, 128, 2), device=dev))
    for scaled in (False, True):
        (po0, pa0), (po1, pa1) = mk(), mk()
        if scaled:
            ops.cac_stats_scaled(Slice(pre2, 64, 64), Slice(pre2, 0, 64), chs, po0, pa0)
            ops.cac_stats_scaled(Slice(p16, 64, 64), Slice(p16, 0, 64), chs, po1, pa1)
        else:
            ops.cac_stats(Slice(pre2, 64, 64), Slice(pre2, 0, 64), po0, pa0)
            ops.cac_stats(Slice(p16, 64, 64), Slice(p16, 0, 64), po1, pa1)
        assert torch.equal(po1[:, 0], po0[:, 0])                             # channel max: exact
        assert rmse(po1[:, 1].cpu(), po0[:, 1].cpu()) < 1e-6                 # channel mean: summation order
        assert torch.equal(pa1[..., 1], pa0[..., 1])                         # per-tile channel maxima: exact
        assert rel_rmse(pa1[..., 0].cpu(), pa0[..., 0].cpu()) < 1e-5
    ch = torch.rand((B, 64), device=dev)
    sp = torch.rand((B, 1, H, W), device=dev)
    oc0 = torch.empty((B, 128, H, W), device=dev)
    oc1 = ops.new_act(B, 128, H, W, dtype, dev)
    ops.cac_apply(Slice(pre2, 0, 64), Slice(pre2, 64, 64), ch, sp, Slice(in2, 0, 64), Slice(in2, 64, 64),
                  Slice(oc0, 0, 64), Slice(oc0, 64, 64))
    ops.cac_apply(Slice(p16, 0, 64), Slice(p16, 64, 64), ch, sp, Slice(i16, 0, 64), Slice(i16, 64, 64),
                  Slice(oc1, 0, 64), Slice(oc1, 64, 64))
    assert torch.equal(ops.to_nchw(oc1).float(), oc0.to(dtype).float())
    y0, y1 = torch.empty((B, 64, H, W), device=dev), ops.new_act(B, 64, H, W, dtype, dev)
    ops.ew_sq_scale(Slice(pre2, 64, 64), ch, Slice(y0))
    ops.ew_sq_scale(Slice(p16, 64, 64), ch, Slice(y1))
    assert torch.equal(ops.to_nchw(y1).float(), y0.to(dtype).float())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 1, 1), (1, 33, 70)])
def test_ew_add_mask(shape, dtype):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    q = lambda seed: _rand((B, 128, H, W), seed).to(dtype).float().to(dev)
    for (C, use_src, use_mask, acc) in [(64, True, False, True), (128, False, True, True), (64, True, True, False)]:
        d0, s0, m0 = q(1), q(2), q(3)
        d1, s1, m1 = (ops.from_nchw(t, dtype) for t in (d0, s0, m0))
        ops.ew_add_mask(Slice(d0, 128 - C, C), Slice(s0, 0, C) if use_src else None, Slice(m0, 0, C) if use_mask else None,
                        accumulate=acc)
        ops.ew_add_mask(Slice(d1, 128 - C, C), Slice(s1, 0, C) if use_src else None, Slice(m1, 0, C) if use_mask else None,
                        accumulate=acc)
        assert torch.equal(ops.to_nchw(d1).float(), d0.to(dtype).float())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("hw", [(480, 640), (470, 627)])
def test_resident_filter_conv3x3_equals_the_staged_kernel(hw, dtype):
    """Round 4: the plain conv3x3 64->64 of a LARGE launch runs as a persistent kernel with the whole filter resident in LDS
    (16 waves, 32 x 32 tiles, one workgroup per CU, one barrier per chunk); small launches keep the staged one-tile kernel.
    Same MFMAs on the same operands in the same order: a batch of 8 (resident form) must equal its four batches of 2 (staged
    form) bit for bit -- forward with ReLU, and the dgrad epilogues (ReLU mask, accumulate, mask over the sum); ragged image
    sizes put partial tiles on both edges."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    H, W = hw
    B = 8
    x = ops.from_nchw(torch.relu(_rand((B, 64, H, W), 1)).to(dev), dtype)
    act = ops.from_nchw(_rand((B, 64, H, W), 4).to(dev), dtype)
    prev = ops.from_nchw(_rand((B, 64, H, W), 5).to(dev), dtype)
    w = _rand((64, 64, 3, 3), 2, scale=(2.0 / (9 * 64)) ** 0.5).to(dev)
    wf, wd = ops.packed_weight(w, L.PACK_FWD, dtype), ops.packed_weight(w, L.PACK_DGRAD, dtype)
    variants = [dict(w=wf, kw=dict(relu=True), init=None),
                dict(w=wd, kw=dict(relu_mask=True), init=None),
                dict(w=wd, kw=dict(accumulate=True), init=prev),
                dict(w=wd, kw=dict(accumulate=True, relu_mask=True, mask_sum=True), init=prev)]
    for v in variants:
        def run(lo, hi):
            xs = x[lo:hi].contiguous()
            y = (v["init"][lo:hi].clone() if v["init"] is not None else ops.new_act(hi - lo, 64, H, W, dtype, dev).fill_(float("nan")))
            kw = dict(v["kw"])
            if kw.get("relu_mask"):
                kw["relu_mask"] = Slice(act[lo:hi].contiguous())
            ops.conv2d(Slice(xs), v["w"], Slice(y), 3, **kw)
            return y
        full = run(0, B)
        parts = torch.cat([run(i, i + 2) for i in range(0, B, 2)], 0)
        assert torch.equal(full, parts), v["kw"]
        assert not torch.isnan(ops.to_nchw(full).float()).any()


@pytest.mark.parametrize("dtype", DT + [torch.float32])
@pytest.mark.parametrize("k", [3, 5])
def test_conv_mask_sum_equals_accumulate_then_mask(k, dtype):
    """CODON_CONV_MASK_SUM (round 4): y = mask > 0 ? conv + y : 0 in the epilogue of the LAST gradient that fans into a
    ReLU output == the accumulating conv followed by a mask pass (codon_ew_add_mask), bit for bit."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = 2, 21, 37
    conv = (lambda t: ops.from_nchw(t, dtype)) if dtype != torch.float32 else (lambda t: t.clone())
    q = lambda c, seed: _rand((B, c, H, W), seed).to(dtype).float().to(dev)
    w = _rand((64, 64, k, k), 2, scale=(2.0 / (k * k * 64)) ** 0.5).to(dev)
    wp = ops.packed_weight(w, L.PACK_DGRAD, dtype)
    gy, prev, act = conv(q(64, 3)), q(128, 5), conv(q(128, 4))
    y0, y1 = conv(prev), conv(prev)
    ops.conv2d(Slice(gy), wp, Slice(y0, 64, 64), k, accumulate=True)
    ops.ew_add_mask(Slice(y0, 64, 64), None, mask=Slice(act, 0, 64))
    ops.conv2d(Slice(gy), wp, Slice(y1, 64, 64), k, accumulate=True, relu_mask=Slice(act, 0, 64), mask_sum=True)
    assert torch.equal(y1, y0)
    frac = float((ops.to_nchw(y1)[:, 64:] == 0).float().mean())
    assert 0.3 < frac < 0.7                                           # the mask did something
    # ... and against torch: the input gradient of a conv (= transposed conv of gy) added to what was there, then masked
    to_f = (lambda t: ops.to_nchw(t).float()) if dtype != torch.float32 else (lambda t: t)
    ref = F.conv_transpose2d(to_f(gy), w.to(dtype).float(), padding=k // 2) + prev[:, 64:]
    ref = torch.where(to_f(act)[:, :64] > 0, ref, torch.zeros_like(ref))
    assert rel_rmse(to_f(y1)[:, 64:].cpu(), ref.cpu()) < _tol(dtype)
    assert torch.equal(to_f(y1)[:, :64], prev[:, :64])                # the other half of the buffer is untouched


@pytest.mark.parametrize("dtype", DT + [torch.float32])
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 1, 1), (1, 33, 70)])
def test_ew_sum_mask(shape, dtype):
    """dst = mask > 0 ? s0 + s1 + s2 + s3 : 0 (round 4: dL/d(fuse) collected in one pass): fp32 sum in source order, rounded
    once to the tensor's dtype; slices of wider buffers; 1..4 sources; with and without the mask."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    q = lambda seed: _rand((B, 128, H, W), seed).to(dtype).float().to(dev)
    srcs_f = [q(11), q(12), q(13), q(14)]
    mask_f = q(15)
    conv = (lambda t: ops.from_nchw(t, dtype)) if dtype != torch.float32 else (lambda t: t.clone())
    srcs, mask = [conv(t) for t in srcs_f], conv(mask_f)
    for n in (1, 2, 3, 4):
        for use_mask in (False, True):
            dst = ops.new_act(B, 128, H, W, dtype, dev).fill_(float("nan"))
            ops.ew_sum_mask(Slice(dst, 64, 64), [Slice(t, 0, 64) for t in srcs[:n]], mask=Slice(mask, 32, 64) if use_mask else None)
            ref = srcs_f[0][:, :64].clone()
            for t in srcs_f[1:n]:
                ref = ref + t[:, :64]
            if use_mask:
                ref = torch.where(mask_f[:, 32:96] > 0, ref, torch.zeros_like(ref))
            got = ops.to_nchw(dst).float()
            assert torch.equal(got[:, 64:], ref.to(dtype).float()), (n, use_mask)
            assert torch.isnan(got[:, :64]).all()


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 1, 1), (1, 50, 70)])
@pytest.mark.parametrize("accumulate_in", [False, True])
def test_cac_backward(shape, dtype, accumulate_in):
    """All four CAC backward launches on blocked tensors vs the fp32 kernels on the same 16-bit-representable values:
    arg-max routing of both max-pools and of the channel max must pick the same elements (ties included: values are
    16-bit, so exact ties across channels and pixels are common)."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    q = lambda seed, s=1.0: _rand((B, 128, H, W), seed, s).to(dtype).float().to(dev)
    g_oc, pre2, g_in = q(1, 0.1), q(2), q(3, 0.1)
    w1, b1, w2 = _rand((8, 128), 7, 0.1).to(dev), _rand((8,), 8, 0.1).to(dev), _rand((64, 8), 9, 0.3).to(dev)
    b2, ws = _rand((64,), 10, 0.1).to(dev), _rand((1, 2, 5, 5), 11, 0.2).to(dev)
    nt = ops.cac_stats_tiles(H, W)
    pooled, partials = torch.empty((B, 2, H, W), device=dev), torch.empty((B, nt, 128, 2), device=dev)
    ch, sp, pools = torch.empty((B, 64), device=dev), torch.empty((B, 1, H, W), device=dev), torch.empty((B, 2, 128), device=dev)
    ops.cac_stats(Slice(pre2, 64, 64), Slice(pre2, 0, 64), pooled, partials)
    ops.cac_gate(B, H, W, partials, w1, b1, w2, b2, ch, pools)
    ops.cac_spatial(pooled, ws, sp)

    def run(conv):
        go, pr, gi = conv(g_oc), conv(pre2), conv(g_in)
        gp = torch.full_like(go, float("nan"))
        outs = ops.cac_backward(Slice(go, 0, 64), Slice(go, 64, 64), Slice(pr, 0, 64), Slice(pr, 64, 64), ch, sp, pooled,
                                pools, w1, b1, w2, ws, Slice(gp, 0, 64), Slice(gp, 64, 64), Slice(gi, 0, 64),
                                Slice(gi, 64, 64), accumulate_in=accumulate_in)
        return outs, ops.to_nchw(gp).float(), ops.to_nchw(gi).float()

    outs0, gp0, gi0 = run(lambda t: t.clone())
    outs1, gp1, gi1 = run(lambda t: ops.from_nchw(t, dtype))
    for a, b in zip(outs1, outs0):                                   # parameter gradients: fp32 either way
        assert rel_rmse(a.cpu(), b.cpu()) < 2e-5
    assert rel_rmse(gp1.cpu(), gp0.cpu()) < _tol(dtype)
    # routing: the few large entries (arg-max pixels / channels) are where a wrong choice would show
    assert float((gp1 - gp0).abs().max()) <= 2 * _tol(dtype) * float(gp0.abs().max())
    assert torch.equal(gi1, gi0.to(dtype).float())


@pytest.mark.parametrize("dtype", DT)
def test_conv_dgrad_and_1x1_against_fp32_kernels(dtype):
    """PACK_DGRAD weights + MASK_RELU / ACCUM_OUT epilogues and the stand-alone 1x1 (both directions), blocked vs fp32."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = 2, 21, 37
    for (k, cin, cout) in [(5, 128, 128), (5, 64, 64), (3, 64, 64), (3, 128, 64), (1, 128, 64)]:
        w = _rand((cout, cin, k, k), 2, scale=(2.0 / (k * k * cout)) ** 0.5).to(dtype).float().to(dev)
        gy = _rand((B, cout, H, W), 3).to(dtype).float().to(dev)
        act = _rand((B, cin, H, W), 4).to(dtype).float().to(dev)
        prev = _rand((B, cin, H, W), 5).to(dtype).float().to(dev)
        g0 = prev.clone()
        ops.conv2d(Slice(gy), ops.packed_weight(w, L.PACK_DGRAD), Slice(g0), k, relu_mask=Slice(act), accumulate=True)
        g1 = ops.from_nchw(prev, dtype)
        ops.conv2d(Slice(ops.from_nchw(gy, dtype)), ops.packed_weight(w, L.PACK_DGRAD, dtype), Slice(g1), k,
                   relu_mask=Slice(ops.from_nchw(act, dtype)), accumulate=True)
        assert rel_rmse(ops.to_nchw(g1).float().cpu(), g0.cpu()) < _tol(dtype), (k, cin, cout)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 1, 1), (1, 33, 70), (2, 64, 40), (1, 8, 32)])
def test_fused_statistics_equal_a_pass_over_the_tensor(shape, dtype):
    """conv_chain1x1(stats=...) + cac_fused_finish + cac_gate_folded (statistics from the conv epilogue) against
    cac_stats + cac_gate (a pass over the stored tensor): both max pools bit for bit, the sums to fp32 summation order,
    and the tensors the conv writes identical with and without the statistics."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    nan = lambda c: ops.new_act(B, c, H, W, dtype, dev).fill_(float("nan"))
    w1, b1, w2, b2 = _rand((8, 128), 7, 0.1).to(dev), _rand((8,), 8, 0.1).to(dev), _rand((64, 8), 9, 0.3).to(dev), _rand((64,), 10, 0.1).to(dev)
    pre2, ref2 = nan(128), nan(128)
    nt = ops.cac_fused_tiles(H, W)
    fz = dict(dtype=torch.float32, device=dev)
    partials = torch.full((B, nt, 128, 2), float("nan"), **fz)
    pool = {0: torch.full((B, 2, H, W), float("nan"), **fz), 64: torch.full((B, 2, H, W), float("nan"), **fz)}
    for choff, seed in ((64, 1), (0, 2)):          # depth stream -> pre2[:, :64], colour stream -> pre2[:, 64:]
        x = ops.from_nchw(torch.relu(_rand((B, 128, H, W), seed)).to(dev), dtype)
        w5 = _rand((128, 128, 5, 5), 10 + seed, (2.0 / (25 * 128)) ** 0.5).to(dev)
        wc = _rand((64, 128, 1, 1), 20 + seed, 0.15).to(dev)
        wp, wcp = ops.packed_weight(w5, L.PACK_FWD, dtype), ops.packed_weight(wc, L.PACK_CHAIN1X1, dtype)
        dst = 0 if choff == 64 else 64
        ops.conv_chain1x1(Slice(x), wp, wcp, Slice(pre2, dst, 64), stats=(pool[choff], partials, choff))
        ops.conv_chain1x1(Slice(x), wp, wcp, Slice(ref2, dst, 64))
    assert torch.equal(pre2, ref2) and not torch.isnan(ops.to_nchw(pre2).float()).any()
    assert not torch.isnan(partials).any() and not torch.isnan(pool[0]).any() and not torch.isnan(pool[64]).any()
    folded, pooled = torch.empty((B, L.CAC_FOLDS, 128, 2), **fz), torch.empty((B, 2, H, W), **fz)
    ch, pools = torch.empty((B, 64), **fz), torch.empty((B, 2, 128), **fz)
    ops.cac_fused_finish(B, H, W, partials, pool[0], pool[64], folded, pooled)
    ops.cac_gate_folded(B, H, W, folded, w1, b1, w2, b2, ch, pools)
    # the same quantities from a pass over the stored tensor
    nt0 = ops.cac_stats_tiles(H, W)
    pooled0, partials0 = torch.empty((B, 2, H, W), **fz), torch.empty((B, nt0, 128, 2), **fz)
    ch0, pools0 = torch.empty((B, 64), **fz), torch.empty((B, 2, 128), **fz)
    ops.cac_stats(Slice(pre2, 64, 64), Slice(pre2, 0, 64), pooled0, partials0)
    ops.cac_gate(B, H, W, partials0, w1, b1, w2, b2, ch0, pools0)
    assert torch.equal(pooled[:, 0], pooled0[:, 0])                      # channel max
    assert torch.equal(pools[:, 1], pools0[:, 1])                        # global max pool
    scale = float(pooled0[:, 1].abs().max()) + 1e-30
    assert float((pooled[:, 1] - pooled0[:, 1]).abs().max()) <= 2e-6 * scale
    assert float((pools[:, 0] - pools0[:, 0]).abs().max()) <= 2e-6 * (float(pools0[:, 0].abs().max()) + 1e-30)
    assert float((ch - ch0).abs().max()) <= 1e-6
    # torch restatement of the pools on the stored values
    F2 = ops.to_nchw(pre2).float()
    Fcat = torch.cat((F2[:, 64:], F2[:, :64]), 1)
    assert torch.equal(pools[:, 1], Fcat.amax((2, 3)))
    assert rel_rmse(pools[:, 0].cpu(), Fcat.double().mean((2, 3)).float().cpu()) < 1e-6


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("kcc", [(5, 64, 64), (3, 64, 64), (3, 128, 64)])
def test_gated_conv_equals_apply_then_conv(kcc, dtype):
    """codon_conv2d_gated_fwd on blocked tensors == cac_apply followed by conv2d, bit for bit (the gate-apply of
    CODON_x4.py:89-91,117-118 formed while the consumer stages its input)."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    k, cin, cout = kcc
    for (B, H, W) in [(2, 19, 45), (1, 1, 1), (1, 33, 70)]:
        pre2 = ops.from_nchw(_rand((B, 128, H, W), 1).to(dev), dtype)
        in2 = ops.from_nchw(torch.relu(_rand((B, 128, H, W), 2)).to(dev), dtype)
        ch, sp = torch.rand((B, 64), device=dev), torch.rand((B, 1, H, W), device=dev)
        w = _rand((cout, cin, k, k), 3, (2.0 / (k * k * cout)) ** 0.5).to(dev)
        wp = ops.packed_weight(w, dtype=dtype)
        oc = ops.new_act(B, 128, H, W, dtype, dev)
        ops.cac_apply(Slice(pre2, 0, 64), Slice(pre2, 64, 64), ch, sp, Slice(in2, 0, 64), Slice(in2, 64, 64),
                      Slice(oc, 0, 64), Slice(oc, 64, 64))
        off = 0 if cin == 128 else 64
        y0, y1 = ops.new_act(B, cout, H, W, dtype, dev), ops.new_act(B, cout, H, W, dtype, dev).fill_(float("nan"))
        ops.conv2d(Slice(oc, off, cin), wp, Slice(y0), k, relu=True)
        ops.conv2d_gated(Slice(pre2, off, cin), Slice(in2, off, cin), ch, sp, wp, Slice(y1), k, relu=True)
        assert torch.equal(y0, y1)
        # emit: the same conv also writes the gated input it staged -- every pixel of every channel exactly once,
        # == cac_apply's output bit for bit, into a slice of a wider buffer whose other channels stay untouched
        y2 = ops.new_act(B, cout, H, W, dtype, dev).fill_(float("nan"))
        xg = ops.new_act(B, 128 + cin, H, W, dtype, dev).fill_(float("nan"))
        ops.conv2d_gated(Slice(pre2, off, cin), Slice(in2, off, cin), ch, sp, wp, Slice(y2), k, relu=True,
                         emit=Slice(xg, 64, cin))
        assert torch.equal(y0, y2)
        got, want = ops.to_nchw(xg), ops.to_nchw(oc)
        assert torch.equal(got[:, 64:64 + cin], want[:, off:off + cin])
        assert torch.isnan(got[:, :64]).all() and torch.isnan(got[:, 64 + cin:]).all()


@pytest.mark.parametrize("dtype", DT + [torch.float32])
def test_inference_schedules_agree_bit_for_bit(dtype):
    """Inference has three schedules for `out*ad_CAC + inputs` (CODON_x4.py:89-91): a cac_apply pass, both sibling
    convs gated, or the conv5x5 gated + emitting and the conv3x3 plain on the emitted tensor (the default).  Same
    arithmetic, same rounding points: identical outputs."""
    import codon_amd
    from codon_amd import model as M
    dev = _dev()
    torch.manual_seed(3)
    net = codon_amd.CODONNet().to(dev).eval()
    if dtype != torch.float32:
        net.set_compute_dtype(dtype)                     # fp32: gated / gated + emit (GATED_16BIT is not consulted)
    x = torch.rand((2, 1, 37, 70), device=dev)
    y = torch.rand((2, 1, 37, 70), device=dev)
    outs = []
    old = (M.GATED_16BIT, M.GATED_EMIT)
    try:
        for g16, emit in ((True, True), (True, False), (False, False)):
            M.GATED_16BIT, M.GATED_EMIT = g16, emit
            with torch.no_grad():
                outs.append(net(x, y).clone())
    finally:
        M.GATED_16BIT, M.GATED_EMIT = old
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_training_schedules_agree_bit_for_bit():
    """bf16 training forward: cac_apply pass vs gated + emitting convs (the emitted tensors are the saved block inputs):
    same output, same gradients, bit for bit."""
    import codon_amd
    from codon_amd import model as M
    dev = _dev()
    torch.manual_seed(5)
    net = codon_amd.CODONNet().to(dev)
    net.set_compute_dtype(torch.bfloat16)
    net.train()
    x = torch.rand((2, 1, 37, 70), device=dev)
    y = torch.rand((2, 1, 37, 70), device=dev)
    gy = torch.randn((2, 1, 37, 70), device=dev)
    res = []
    old = M.GATED_EMIT
    try:
        for emit in (True, False):
            M.GATED_EMIT = emit
            net.zero_grad(set_to_none=True)
            out = net(x, y)
            out.backward(gy)
            res.append((out.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}))
    finally:
        M.GATED_EMIT = old
    assert torch.equal(res[0][0], res[1][0])
    assert res[0][1].keys() == res[1][1].keys() and len(res[0][1]) >= 40
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], res[1][1][n]), n
    # ... and so do the fused / two-kernel backward of the 1x1 convs
    from codon_amd import autograd as A
    oldf = A.FUSED_1X1_BWD
    try:
        A.FUSED_1X1_BWD = not oldf
        net.zero_grad(set_to_none=True)
        out = net(x, y)
        out.backward(gy)
        g2 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        A.FUSED_1X1_BWD = oldf
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], g2[n]), n
    # ... and the CAC backward with / without its apply pass (round 4: cac_backward_fused + conv1x1_bwd_gated)
    oldc = A.FUSED_CAC_BWD
    try:
        A.FUSED_CAC_BWD = not oldc
        net.zero_grad(set_to_none=True)
        out = net(x, y)
        out.backward(gy)
        g3 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        A.FUSED_CAC_BWD = oldc
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], g3[n]), n
    # ... and the ReLU mask of in2 applied in the last dgrad's epilogue or as a pass
    oldm = A.MASK_IN_EPILOGUE
    try:
        A.MASK_IN_EPILOGUE = not oldm
        net.zero_grad(set_to_none=True)
        out = net(x, y)
        out.backward(gy)
        g4 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        A.MASK_IN_EPILOGUE = oldm
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], g4[n]), n
    # ... and dL/d(inputs) += dL/d(out_i) in the last dgrad's epilogue (codon_conv2d_sum_into_fwd) or in the reduce pass
    olds = A.SUM_IN_DGRAD
    try:
        A.SUM_IN_DGRAD = not olds
        net.zero_grad(set_to_none=True)
        out = net(x, y)
        out.backward(gy)
        g5 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        A.SUM_IN_DGRAD = olds
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], g5[n]), n
    # ... and every reduction of the backward in one launch at its end, or a small launch behind each producer (round 5)
    oldd = A.DEFER_REDUCE
    try:
        A.DEFER_REDUCE = not oldd
        net.zero_grad(set_to_none=True)
        out = net(x, y)
        out.backward(gy)
        g6 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        A.DEFER_REDUCE = oldd
    for n in res[0][1]:
        assert torch.equal(res[0][1][n], g6[n]), n
    # ... dL/d(fuse) collected by the trunk's input-gradient convs (round 5) or by ew_sum_mask: NOT the same roundings -- the
    # running sum is stored in bf16 after each term (as the reference's own bf16 autograd accumulates, and as dL/d(inputs)
    # does above), the pass rounded the fp32 sum of four terms once, and the trunk's two input-gradient convs swap places (the
    # conv5x5 runs last and carries the add) -- so everything from the trunk's second iteration on agrees to bf16 rounding
    # noise, what precedes it in the backward (conv11, output) bit for bit
    oldg = A.SUM_GFUSE_IN_DGRAD
    try:
        A.SUM_GFUSE_IN_DGRAD = not oldg
        net.zero_grad(set_to_none=True)
        out = net(x, y)
        out.backward(gy)
        g7 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        A.SUM_GFUSE_IN_DGRAD = oldg
    for n in res[0][1]:
        if n.split(".")[0] in ("conv11", "output"):
            assert torch.equal(res[0][1][n], g7[n]), n
        else:
            a, b_ = res[0][1][n].double(), g7[n].double()
            # (the gate tensors carry less than two digits in bf16 whoever computes them: cancellation; they are
            # held to the reference's own bf16 autograd by test_bf16_gradients_vs_reference_bf16_autograd)
            tol = 0.15 if n.startswith("attention") else 2e-2
            assert float((a - b_).norm() / b_.norm()) < tol, (n, float((a - b_).norm() / b_.norm()))


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 21, 37), (1, 1, 1), (1, 33, 70)])
@pytest.mark.parametrize("accumulate", [False, True])
def test_conv_sum_into_equals_conv_then_add(shape, dtype, accumulate):
    """codon_conv2d_sum_into_fwd: y (+)= conv5x5(x) and total += y in ONE epilogue == codon_conv2d_fwd followed by an
    elementwise add of the STORED y (codon_ew_add_mask), bit for bit, on slices of wider buffers; y also against torch's
    transposed conv (the launch is a dgrad: PACK_DGRAD weights)."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    q = lambda c, seed: _rand((B, c, H, W), seed).to(dtype).float().to(dev)
    w = _rand((64, 64, 5, 5), 2, scale=(2.0 / (25 * 64)) ** 0.5).to(dev)
    wp = ops.packed_weight(w, L.PACK_DGRAD, dtype)
    gy_f, prev_f, tot_f = q(128, 3), q(128, 5), q(192, 6)
    gy = ops.from_nchw(gy_f, dtype)
    y0, y1 = ops.from_nchw(prev_f, dtype), ops.from_nchw(prev_f, dtype)
    t0, t1 = ops.from_nchw(tot_f, dtype), ops.from_nchw(tot_f, dtype)
    ops.conv2d(Slice(gy, 64, 64), wp, Slice(y0, 64, 64), 5, accumulate=accumulate)
    ops.ew_add_mask(Slice(t0, 128, 64), Slice(y0, 64, 64))
    ops.conv2d_sum_into(Slice(gy, 64, 64), wp, Slice(y1, 64, 64), 5, Slice(t1, 128, 64), accumulate=accumulate)
    assert torch.equal(y1, y0) and torch.equal(t1, t0)
    assert torch.equal(ops.to_nchw(t1)[:, :128].float(), tot_f[:, :128])       # the rest of the buffer is untouched
    ref = F.conv_transpose2d(gy_f[:, 64:], w.to(dtype).float(), padding=2) + (prev_f[:, 64:] if accumulate else 0)
    assert rel_rmse(ops.to_nchw(y1)[:, 64:].float().cpu(), ref.cpu()) < _tol(dtype)
    ref_t = tot_f[:, 128:] + ops.to_nchw(y1)[:, 64:].float()
    assert torch.equal(ops.to_nchw(t1)[:, 128:].float(), ref_t.to(dtype).float())


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 1, 1), (1, 33, 70), (2, 64, 96)])
def test_conv1x1_bwd_equals_wgrad_plus_masked_dgrad(shape, dtype):
    """codon_conv1x1_bwd (dW and the ReLU-masked dX of a 128 -> 64 1x1 conv from one pass over x and gy) == codon_conv2d_wgrad
    + codon_conv2d_fwd(PACK_DGRAD, MASK_RELU), bit for bit, including accumulation into dW and slices of wider buffers."""
    from codon_amd import ops
    from codon_amd import _lib as L
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    x = ops.from_nchw(torch.relu(_rand((B, 192, H, W), 1)).to(dev), dtype)       # the conv input = channels 64..191
    g = ops.from_nchw(_rand((B, 128, H, W), 2).to(dev), dtype)                  # gy = channels 64..127
    w = _rand((64, 128, 1, 1), 3, 0.1).to(dev)
    wp = ops.packed_weight(w, mode=L.PACK_DGRAD, dtype=dtype)
    xs, gs = Slice(x, 64, 128), Slice(g, 64, 64)
    dw0 = torch.full((64, 128, 1, 1), 0.5, device=dev)
    dw1 = dw0.clone()
    gx0 = ops.new_act(B, 128, H, W, dtype, dev)
    gx1 = ops.new_act(B, 256, H, W, dtype, dev).fill_(float("nan"))
    for acc in (False, True):
        ops.conv2d_wgrad(xs, gs, dw0, 1, accumulate=acc)
        ops.conv2d(gs, wp, Slice(gx0), 1, relu_mask=xs)
        ops.conv1x1_bwd(xs, gs, wp, Slice(gx1, 64, 128), dw1, accumulate=acc)
        assert torch.equal(dw0, dw1)
        got = ops.to_nchw(gx1)
        assert torch.equal(got[:, 64:192], ops.to_nchw(gx0))
        assert torch.isnan(got[:, :64]).all() and torch.isnan(got[:, 192:]).all()
    ref = torch.einsum("bohw,oi->bihw", ops.to_nchw(g)[:, 64:].float(), w[:, :, 0, 0]) * (ops.to_nchw(x)[:, 64:] > 0)
    assert rel_rmse(ops.to_nchw(gx0).float().cpu(), ref.cpu()) < _tol(dtype)
    # dW against torch in float64 on the same 16-bit values: written once (accumulate=False), then added once more
    dw_ref = 2 * torch.einsum("bohw,bihw->oi", ops.to_nchw(g)[:, 64:].double(), ops.to_nchw(x)[:, 64:].double())
    assert rel_rmse(dw1[:, :, 0, 0].double().cpu(), dw_ref.cpu()) < 2e-5


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 1, 1), (1, 50, 70), (2, 64, 96)])
@pytest.mark.parametrize("accumulate_in", [False, True])
def test_fused_cac_backward_equals_the_apply_pass(shape, dtype, accumulate_in):
    """Round 4: the CAC gate backward without its apply pass.  codon_cac_bwd_reduce_acc (pass A that also records every
    pixel's arg-max channel and folds dL/d(out) into dL/d(inputs)) + codon_conv1x1_bwd_gated (dL/d(pre) formed from dL/d(out)
    while the 1x1 backward stages it) against the four-kernel form + codon_conv1x1_bwd on the g_pre it stores: dL/d(inputs),
    all five parameter gradients, the 1x1 convs' dW and the ReLU-masked dL/d(r2) of BOTH streams bit for bit -- values are
    16-bit, so exact ties in the channel max (routing to the first Fcat channel) and in the global max pools are common."""
    from codon_amd import ops
    from codon_amd import _lib as L
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    # coarse values: many exact ties across channels and pixels
    q = lambda seed, s=1.0: ops.from_nchw(((_rand((B, 128, H, W), seed, s) * 4).round() / 4).to(dev), dtype)
    g_oc, pre2, g_in0 = q(1, 0.5), q(2), q(3, 0.5)
    r2 = {0: ops.from_nchw(torch.relu(_rand((B, 128, H, W), 4)).to(dev), dtype),
          64: ops.from_nchw(torch.relu(_rand((B, 128, H, W), 5)).to(dev), dtype)}
    w1, b1, w2 = _rand((8, 128), 7, 0.1).to(dev), _rand((8,), 8, 0.1).to(dev), _rand((64, 8), 9, 0.3).to(dev)
    b2, ws = _rand((64,), 10, 0.1).to(dev), _rand((1, 2, 5, 5), 11, 0.2).to(dev)
    wc = {0: ops.packed_weight(_rand((64, 128, 1, 1), 12, 0.1).to(dev), mode=L.PACK_DGRAD, dtype=dtype),
          64: ops.packed_weight(_rand((64, 128, 1, 1), 13, 0.1).to(dev), mode=L.PACK_DGRAD, dtype=dtype)}
    nt = ops.cac_stats_tiles(H, W)
    pooled, partials = torch.empty((B, 2, H, W), device=dev), torch.empty((B, nt, 128, 2), device=dev)
    ch, sp, pools = torch.empty((B, 64), device=dev), torch.empty((B, 1, H, W), device=dev), torch.empty((B, 2, 128), device=dev)
    ops.cac_stats(Slice(pre2, 64, 64), Slice(pre2, 0, 64), pooled, partials)
    ops.cac_gate(B, H, W, partials, w1, b1, w2, b2, ch, pools)
    ops.cac_spatial(pooled, ws, sp)
    # the four-kernel form: [depth | colour] halves, depth = Fcat channels 64..127
    gi0, gp0 = g_in0.clone(), ops.new_act(B, 128, H, W, dtype, dev)
    outs0 = ops.cac_backward(Slice(g_oc, 0, 64), Slice(g_oc, 64, 64), Slice(pre2, 0, 64), Slice(pre2, 64, 64), ch, sp, pooled,
                             pools, w1, b1, w2, ws, Slice(gp0, 0, 64), Slice(gp0, 64, 64), Slice(gi0, 0, 64),
                             Slice(gi0, 64, 64), accumulate_in=accumulate_in)
    gi1 = g_in0.clone()
    *outs1, gate = ops.cac_backward_fused(Slice(g_oc, 0, 64), Slice(g_oc, 64, 64), Slice(pre2, 0, 64), Slice(pre2, 64, 64), ch,
                                          sp, pooled, pools, w1, b1, w2, ws, Slice(gi1, 0, 64), Slice(gi1, 64, 64),
                                          accumulate_in=accumulate_in)
    assert torch.equal(gi1, gi0)
    for a, b in zip(outs1, outs0):
        assert torch.equal(a, b)
    # the per-pixel arg-max channel against torch.max over Fcat = cat(colour, depth) (first maximum wins)
    F2 = ops.to_nchw(pre2).float()
    Fcat = torch.cat((F2[:, 64:], F2[:, :64]), 1)
    assert torch.equal(gate["argch"].long(), Fcat.max(dim=1).indices)
    for fbase, coff in ((64, 0), (0, 64)):                # depth stream: g_oc / g_pre channels 0..63; colour: 64..127
        for acc in (False, True):
            dw0 = torch.full((64, 128, 1, 1), 0.25, device=dev)
            dw1 = dw0.clone()
            gx0, gx1 = ops.new_act(B, 128, H, W, dtype, dev), ops.new_act(B, 128, H, W, dtype, dev)
            ops.conv1x1_bwd(Slice(r2[fbase]), Slice(gp0, coff, 64), wc[fbase], Slice(gx0), dw0, accumulate=acc)
            ops.conv1x1_bwd_gated(Slice(r2[fbase]), Slice(g_oc, coff, 64), wc[fbase], Slice(gx1), dw1, gate, fbase, accumulate=acc)
            assert torch.equal(dw1, dw0), (fbase, acc)
            assert torch.equal(gx1, gx0), (fbase, acc)


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 33, 70)])
def test_fused_cac_backward_against_oracle_autograd(shape, dtype):
    """The round-4 entry points (codon_cac_bwd_reduce_acc, codon_conv1x1_bwd_gated) against torch AUTOGRAD through the
    oracle's restatement of the block tail (oracle/codon_oracle.py: cac_channel, cac_spatial, the gate-apply of
    CODON_x4.py:85-91) in float64 -- not against another HIP kernel.  Inputs are 16-bit representable, `pre` is a leaf on
    both sides (so the arg-max routing of the three max-pools sees identical values), the loss is <g_out, [out | out_c]>:
        dL/d(inputs)  = g_out (+ what was there);   the five CAC parameter gradients;
        dL/d(pre)     -> dW = sum_p g_pre (x) r2   and   dL/d(r2) = W^T g_pre * [r2 > 0]   of both 1x1 convs."""
    from codon_amd import ops
    from codon_amd import _lib as L
    from codon_amd.ops import Slice
    from oracle import codon_oracle as orc
    dev = _dev()
    B, H, W = shape
    q = lambda seed, s=1.0: _rand((B, 128, H, W), seed, s).to(dtype).float()
    g_oc, pre2, g_in0 = q(1, 0.5), q(2), q(3, 0.5)                 # [depth | colour] halves, as the product lays them out
    r2 = {64: torch.relu(q(4)), 0: torch.relu(q(5))}               # inputs of confuse (depth, Fcat 64..) / confuse_c (colour)
    w1, b1, w2 = _rand((8, 128), 7, 0.1), _rand((8,), 8, 0.1), _rand((64, 8), 9, 0.3)
    b2, ws = _rand((64,), 10, 0.1), _rand((1, 2, 5, 5), 11, 0.2)
    wconv = {64: _rand((64, 128, 1, 1), 12, 0.1).to(dtype).float(), 0: _rand((64, 128, 1, 1), 13, 0.1).to(dtype).float()}

    # ---- oracle: float64 autograd
    d64 = lambda t: t.double().clone().requires_grad_(True)
    pre_l, w1_l, b1_l, w2_l, b2_l, ws_l = d64(pre2), d64(w1), d64(b1), d64(w2), d64(b2), d64(ws)
    pre_d, pre_c = pre_l[:, :64], pre_l[:, 64:]
    Fcat = torch.cat((pre_c, pre_d), 1)                                          # :85  colour | depth
    g = orc.cac_channel(Fcat, w1_l, b1_l, w2_l, b2_l)[:, :, None, None] * orc.cac_spatial(Fcat, ws_l)   # :86-89
    out = torch.cat((pre_d * g, pre_c * g), 1)                                    # :90-91 without the `+ inputs` leaf
    (out * g_oc.double()).sum().backward()
    g_pre = pre_l.grad                                                           # [depth | colour]
    ref_params = [w1_l.grad, b1_l.grad, w2_l.grad, b2_l.grad, ws_l.grad]

    # ---- HIP: forward statistics / gates (fp32 kernels on the same values), then the fused backward
    gd = lambda t: ops.from_nchw(t.to(dev), dtype)
    P2, G, GI = gd(pre2), gd(g_oc), gd(g_in0)
    pd = lambda t: t.to(dev)
    nt = ops.cac_stats_tiles(H, W)
    pooled, partials = torch.empty((B, 2, H, W), device=dev), torch.empty((B, nt, 128, 2), device=dev)
    ch, sp, pools = torch.empty((B, 64), device=dev), torch.empty((B, 1, H, W), device=dev), torch.empty((B, 2, 128), device=dev)
    ops.cac_stats(Slice(P2, 64, 64), Slice(P2, 0, 64), pooled, partials)
    ops.cac_gate(B, H, W, partials, pd(w1), pd(b1), pd(w2), pd(b2), ch, pools)
    ops.cac_spatial(pooled, pd(ws), sp)
    *outs, gate = ops.cac_backward_fused(Slice(G, 0, 64), Slice(G, 64, 64), Slice(P2, 0, 64), Slice(P2, 64, 64), ch, sp, pooled,
                                         pools, pd(w1), pd(b1), pd(w2), pd(ws), Slice(GI, 0, 64), Slice(GI, 64, 64),
                                         accumulate_in=True)
    # dL/d(inputs): out = pre * g + inputs  ->  the upstream gradient, added to what the buffer held (one 16-bit rounding)
    assert torch.equal(ops.to_nchw(GI).float().cpu(), (g_in0 + g_oc).to(dtype).float())
    for name, a, b in zip(("mlp.1.weight", "mlp.1.bias", "mlp.3.weight", "mlp.3.bias", "spatial.conv.weight"), outs, ref_params):
        assert rel_rmse(a.cpu().double(), b) < 1e-4, name
    for fbase, coff in ((64, 0), (0, 64)):
        wp = ops.packed_weight(wconv[fbase].to(dev), mode=L.PACK_DGRAD, dtype=dtype)
        dw = torch.zeros((64, 128, 1, 1), device=dev)
        gx = ops.new_act(B, 128, H, W, dtype, dev)
        ops.conv1x1_bwd_gated(Slice(gd(r2[fbase])), Slice(G, coff, 64), wp, Slice(gx), dw, gate, fbase)
        gp = g_pre[:, coff:coff + 64]
        dw_ref = torch.einsum("bohw,bihw->oi", gp, r2[fbase].double())
        gx_ref = torch.einsum("bohw,oi->bihw", gp, wconv[fbase][:, :, 0, 0].double()) * (r2[fbase] > 0)
        got = ops.to_nchw(gx).float().cpu().double()
        assert rel_rmse(dw[:, :, 0, 0].cpu().double(), dw_ref) < _tol(dtype), fbase
        assert rel_rmse(got, gx_ref) < _tol(dtype), fbase
        # the routed terms are the few large entries of dL/d(pre): a wrong arg-max would show as an O(1) outlier here
        assert float((got - gx_ref).abs().max()) <= 4 * _tol(dtype) * float(gx_ref.abs().max()) + 1e-6, fbase


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_two_stream_schedule_is_bit_identical(dtype):
    """Small grids (one 128 x 128 image: BASELINE configs[0]) run the depth and the colour stream of a block on two HIP
    streams; same kernels, same operands: the output must not change, eager and under hipGraph capture."""
    import codon_amd
    from codon_amd import model as M
    dev = _dev()
    torch.manual_seed(7)
    net = codon_amd.CODONNet().to(dev).eval()
    if dtype != torch.float32:
        net.set_compute_dtype(dtype)
    x = torch.rand((1, 1, 128, 128), device=dev)
    y = torch.rand((1, 1, 128, 128), device=dev)
    old = M.TWO_STREAMS
    outs = []
    try:
        for two in (True, False):
            M.TWO_STREAMS = two
            with torch.no_grad():
                for _ in range(3):                       # repeated: a missing join would show as a race
                    outs.append(net(x, y).clone())
        M.TWO_STREAMS = True
        from codon_amd.graph import GraphedCODON
        g = GraphedCODON(net, x, y)
        outs.append(g(x, y).clone())
        outs.append(g(x, y).clone())
    finally:
        M.TWO_STREAMS = old
    torch.cuda.synchronize()
    assert torch.isfinite(outs[0]).all()
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("shape", [(1, 37, 70), (3, 64, 96), (2, 5, 3), (1, 370, 463)])
@pytest.mark.parametrize("fused", [True, False])
def test_cac_tail_equals_the_separate_launches(shape, fused):
    """codon_cac_tail_fwd (round 5: the whole gate of a block in one launch) against the launches it replaces, bit for bit:
    fused = the 16-bit path's operands (two per-stream {max, sum} maps + per-conv-tile partials -> fold, combine, gate,
    spatial); not fused = the fp32 path's (pooled + per-stats-tile partials, <= CODON_CAC_FOLDS tiles -> gate, spatial).
    The arrival counters must come back as zeros, and a second launch on the same buffers must give the same bits."""
    from codon_amd import _lib as L, ops
    dev = _dev()
    B, H, W = shape
    nt = ops.cac_fused_tiles(H, W) if fused else ops.cac_stats_tiles(H, W)
    if not fused and nt > L.CAC_FOLDS:
        pytest.skip("fp32 path: the one-launch form is used up to CODON_CAC_FOLDS tiles")
    g = torch.Generator().manual_seed(B * 1000 + H)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    partials = rnd(B, nt, 128, 2)
    w1, b1, w2, b2, ws = rnd(8, 128) * 0.1, rnd(8) * 0.1, rnd(64, 8) * 0.3, rnd(64) * 0.1, rnd(1, 2, 5, 5) * 0.2
    mk = lambda: (torch.empty((B, 64), device=dev), torch.empty((B, 1, H, W), device=dev), torch.empty((B, 2, 128), device=dev))
    (ch0, sp0, po0), (ch1, sp1, po1) = mk(), mk()
    folded = torch.empty((B, L.CAC_FOLDS, 128, 2), device=dev)
    counters = torch.zeros((B,), dtype=torch.int32, device=dev)
    if fused:
        pool_c, pool_d = rnd(B, 2, H, W), rnd(B, 2, H, W)
        pooled0 = torch.empty((B, 2, H, W), device=dev)
        ops.cac_fused_finish(B, H, W, partials, pool_c, pool_d, folded, pooled0)
        ops.cac_gate_folded(B, H, W, folded, w1, b1, w2, b2, ch0, po0)
        ops.cac_spatial(pooled0, ws, sp0)
        for pooled1 in (torch.full((B, 2, H, W), float("nan"), device=dev), None):       # training keeps pooled, inference does not
            folded.fill_(float("nan"))
            ops.cac_tail(B, H, W, partials, pool_c, pool_d, pooled1, folded, counters, w1, b1, w2, b2, ws, ch1, sp1, po1)
            assert torch.equal(ch1, ch0) and torch.equal(sp1, sp0) and torch.equal(po1, po0)
            assert pooled1 is None or torch.equal(pooled1, pooled0)
            assert int(counters.abs().sum()) == 0
    else:
        pooled = rnd(B, 2, H, W)
        ops.cac_gate(B, H, W, partials, w1, b1, w2, b2, ch0, po0)
        ops.cac_spatial(pooled, ws, sp0)
        for _ in range(2):
            ops.cac_tail(B, H, W, partials, None, None, pooled, folded, counters, w1, b1, w2, b2, ws, ch1, sp1, po1)
            assert torch.equal(ch1, ch0) and torch.equal(sp1, sp0) and torch.equal(po1, po0)
            assert int(counters.abs().sum()) == 0


@pytest.mark.parametrize("shape", [(1, 37, 70), (3, 64, 96), (2, 5, 3), (1, 370, 463)])
@pytest.mark.parametrize("fused", [True, False])
def test_cac_tail_against_the_oracle_gates(shape, fused):
    """Round 6: codon_cac_tail_fwd's own oracle anchor (the bit-identity test above compares it with older HIP launches
    only).  A random Fcat (B,128,H,W) is reduced on the CPU to exactly the operands the launch takes -- per-tile {sum, max}
    partials (the pixels of an image cut into ntiles groups: the launch folds groups, their geometry is the producer's
    business), the two per-stream {max, SUM} maps or the finished {max, mean} map -- and the launch's ch / sp must be
    oracle.cac_channel / oracle.cac_spatial of that Fcat (/root/reference/CODON_X4/CAC_module.py:38-63, 78-94)."""
    from codon_amd import _lib as L, ops
    from oracle import codon_oracle as orc
    dev = _dev()
    B, H, W = shape
    nt = ops.cac_fused_tiles(H, W) if fused else ops.cac_stats_tiles(H, W)
    if not fused and nt > L.CAC_FOLDS:
        pytest.skip("fp32 path: the one-launch form is used up to CODON_CAC_FOLDS tiles")
    assert nt <= H * W
    g = torch.Generator().manual_seed(B * 77 + H)
    fcat = torch.randn(B, 128, H, W, generator=g) * 1.5 + 0.3            # channels 0..63 colour, 64..127 depth (CODON_x4.py:85)
    w1, b1 = torch.randn(8, 128, generator=g) * 0.1, torch.randn(8, generator=g) * 0.1
    w2, b2 = torch.randn(64, 8, generator=g) * 0.3, torch.randn(64, generator=g) * 0.1
    ws = torch.randn(1, 2, 5, 5, generator=g) * 0.2
    with torch.no_grad():
        want_ch = orc.cac_channel(fcat, w1, b1, w2, b2)
        want_sp = orc.cac_spatial(fcat, ws)
    flat = fcat.reshape(B, 128, H * W)
    cuts = np.linspace(0, H * W, nt + 1).astype(np.int64)
    partials = torch.empty(B, nt, 128, 2)
    for t in range(nt):
        seg = flat[:, :, int(cuts[t]):int(cuts[t + 1])]
        partials[:, t, :, 0] = seg.sum(2)
        partials[:, t, :, 1] = seg.max(2)[0]
    stream_map = lambda f: torch.stack((f.max(1)[0], f.sum(1)), 1).contiguous()
    pooled_ref = torch.stack((fcat.max(1)[0], fcat.mean(1)), 1).contiguous()
    ch, sp = torch.empty((B, 64), device=dev), torch.empty((B, 1, H, W), device=dev)
    pools = torch.empty((B, 2, 128), device=dev)
    folded = torch.empty((B, L.CAC_FOLDS, 128, 2), device=dev)
    counters = torch.zeros((B,), dtype=torch.int32, device=dev)
    cu = lambda *ts: [t.to(dev) for t in ts]
    if fused:
        pooled = torch.full((B, 2, H, W), float("nan"), device=dev)
        ops.cac_tail(B, H, W, partials.to(dev), stream_map(fcat[:, :64]).to(dev), stream_map(fcat[:, 64:]).to(dev), pooled, folded,
                     counters, *cu(w1, b1, w2, b2, ws), ch, sp, pools)
        assert torch.equal(pooled[:, 0].cpu(), pooled_ref[:, 0])             # the channel max is exact
        assert rel_rmse(pooled[:, 1].cpu(), pooled_ref[:, 1]) <= 1e-6
    else:
        ops.cac_tail(B, H, W, partials.to(dev), None, None, pooled_ref.to(dev), folded, counters, *cu(w1, b1, w2, b2, ws), ch, sp, pools)
    torch.cuda.synchronize()
    assert int(counters.abs().sum()) == 0
    assert rel_rmse(ch.cpu(), want_ch) <= 1e-5 and float((ch.cpu() - want_ch).abs().max()) <= 1e-5
    assert rel_rmse(sp.cpu(), want_sp) <= 1e-5 and float((sp.cpu() - want_sp).abs().max()) <= 1e-5


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_small_parameters_as_one_flat_fp32_buffer(dtype):
    """ops.params_f32 (codon_cast_multi): 28 tensors of a model cast to 16 bits -> fp32 in one launch, exact."""
    from codon_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    shapes = [(64, 1, 3, 3), (64, 1, 3, 3), (1, 64, 3, 3)] + [(8, 128), (8,), (64, 8), (64,), (1, 2, 5, 5)] * 5 + [(3,)] * 9
    ts = [torch.randn(s, generator=g).to(dtype).to(dev) for s in shapes]          # 37 tensors: more than one launch's worth
    out = ops.params_f32(ts)
    assert len(out) == len(ts)
    for o, t in zip(out, ts):
        assert o.dtype == torch.float32 and o.shape == t.shape and torch.equal(o, t.float())
    if dtype == torch.float32:
        assert all(o.data_ptr() == t.data_ptr() for o, t in zip(out, ts))        # nothing to convert: the parameters themselves


@pytest.mark.parametrize("mode", ["fp16", "bf16", "fp32"])
def test_forward_with_one_launch_gate_and_two_streams_is_bit_identical(mode):
    """The schedule switches of round 5 change launches, not arithmetic: the one-launch gate (CAC_TAIL) and the two-stream
    schedule at a grid between the old and the new threshold (TWO_STREAMS_MAX16) leave every output bit in place."""
    import codon_amd
    from codon_amd import model as M
    dev = _dev()
    torch.manual_seed(3)
    net = codon_amd.CODONNet().to(dev).eval()
    if mode == "fp16":
        net = net.half()
    elif mode == "bf16":
        net.set_compute_dtype(torch.bfloat16)
    H, W = (96, 128) if mode != "fp32" else (64, 96)          # fp32: 6 stats tiles <= CODON_CAC_FOLDS
    x = torch.rand((1, 1, H, W), device=dev)
    y = torch.rand((1, 1, H, W), device=dev)
    if mode == "fp16":
        x, y = x.half(), y.half()
    old = (M.CAC_TAIL, M.TWO_STREAMS_MAX16, M.TWO_STREAMS_MAX32, M.PAIR_MAX16, M.PAIR_MAX32)
    outs = {False: [], True: []}
    try:
        for tail, m16, m32, pr in ((False, 0, 0, 0), (True, 0, 0, 0), (True, 4096, 4096, 0), (False, 4096, 4096, 0),
                                   (True, 0, 0, 4096), (False, 4096, 4096, 4096)):
            M.CAC_TAIL, M.TWO_STREAMS_MAX16, M.TWO_STREAMS_MAX32, M.PAIR_MAX16, M.PAIR_MAX32 = tail, m16, m32, pr, pr
            with torch.no_grad():
                outs[tail].append(net(x, y).clone())
            torch.cuda.synchronize()
    finally:
        M.CAC_TAIL, M.TWO_STREAMS_MAX16, M.TWO_STREAMS_MAX32, M.PAIR_MAX16, M.PAIR_MAX32 = old
    for grp in outs.values():
        assert all(torch.equal(grp[0], o) for o in grp[1:])
    if mode == "fp32":
        # 24 small statistics tiles: the one-launch gate folds them in pairs before finishing the pools, the separate gate
        # kernel adds them one by one -- fp32 re-association of a 24-term sum, nothing else
        assert rel_rmse(outs[True][0].float().cpu(), outs[False][0].float().cpu()) < 2e-5     # (measured 1.5e-6)
    else:
        assert torch.equal(outs[True][0], outs[False][0])


@pytest.mark.parametrize("dtype", DT)
def test_conv_pair_is_one_launch_and_bit_identical(dtype):
    """ops.conv_pair (codon_conv_pair_begin / _end): two independent convs of one shape leave as ONE launch with the bits of
    two launches -- plain, chained 1x1 with statistics, gated + emitting; shapes / kernels without a common form fall back to
    two launches (the 5x5 | 3x3 64->64 couple has one since round 6: test_mix53_conv5x5_and_conv3x3_as_one_grid); a bracket is
    per thread and cannot nest."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = 1, 37, 70
    q = lambda c, seed: ops.from_nchw(_rand((B, c, H, W), seed).to(dtype).float().to(dev), dtype)
    xa, xb = q(128, 1), q(128, 2)
    wt = lambda co, ci, k, seed: ops.packed_weight(_rand((co, ci, k, k), seed, (2.0 / (k * k * co)) ** 0.5).to(dev), L.PACK_FWD, dtype)
    w5a, w5b, w3a, w3b = wt(64, 64, 5, 3), wt(64, 64, 5, 4), wt(64, 64, 3, 5), wt(64, 64, 3, 6)
    w3c = wt(64, 128, 3, 9)

    def run(paired):
        outs = [ops.new_act(B, 128, H, W, dtype, dev).zero_() for _ in range(4)]
        with ops.conv_pair(dev, paired) as pr:
            ops.conv2d(Slice(xa, 0, 64), w5a, Slice(outs[0], 64, 64), 5, relu=True)
            ops.conv2d(Slice(xb, 64, 64), w5b, Slice(outs[1], 0, 64), 5, relu=True)
        n1 = pr.launches
        with ops.conv_pair(dev, paired) as pr:             # a 5x5 64->64 and a 3x3 128->64: no common form -> two launches
            ops.conv2d(Slice(xa, 0, 64), w5a, Slice(outs[2], 0, 64), 5)
            ops.conv2d(Slice(xb), w3c, Slice(outs[2], 64, 64), 3)
        n2 = pr.launches
        with ops.conv_pair(dev, paired) as pr:
            ops.conv2d(Slice(xa, 64, 64), w3a, Slice(outs[3], 0, 64), 3, relu=True)
            ops.conv2d(Slice(xb, 0, 64), w3b, Slice(outs[3], 64, 64), 3, relu=True)
        return outs, (n1, n2, pr.launches)

    o0, _ = run(False)
    o1, n = run(True)
    assert n == (1, 2, 1), n
    assert all(torch.equal(a, b) for a, b in zip(o0, o1))
    # chained 1x1 + statistics, and gated + emitting, as pairs
    w5 = wt(128, 128, 5, 7)
    w1 = ops.packed_weight(_rand((64, 128, 1, 1), 8, 0.1).to(dev), L.PACK_CHAIN1X1, dtype)
    ch, sp = torch.rand((B, 64), device=dev), torch.rand((B, 1, H, W), device=dev)
    nt = ops.cac_fused_tiles(H, W)

    def run2(paired):
        pre = ops.new_act(B, 128, H, W, dtype, dev).zero_()
        pc, pd = torch.zeros((B, 2, H, W), device=dev), torch.zeros((B, 2, H, W), device=dev)
        part = torch.zeros((B, nt, 128, 2), device=dev)
        with ops.conv_pair(dev, paired) as pr:
            ops.conv_chain1x1(Slice(xa), w5, w1, Slice(pre, 64, 64), stats=(pc, part, 0))
            ops.conv_chain1x1(Slice(xb), w5, w1, Slice(pre, 0, 64), stats=(pd, part, 64))
        n1 = pr.launches
        y, em = ops.new_act(B, 128, H, W, dtype, dev).zero_(), ops.new_act(B, 128, H, W, dtype, dev).zero_()
        with ops.conv_pair(dev, paired) as pr:
            ops.conv2d_gated(Slice(pre, 64, 64), Slice(xa, 64, 64), ch, sp, w5a, Slice(y, 0, 64), 5, relu=True, emit=Slice(em, 64, 64))
            ops.conv2d_gated(Slice(pre, 0, 64), Slice(xa, 0, 64), ch, sp, w5b, Slice(y, 64, 64), 5, relu=True, emit=Slice(em, 0, 64))
        return (pre, pc, pd, part, y, em), (n1, pr.launches)

    r0, _ = run2(False)
    r1, n = run2(True)
    assert n == (1, 1), n
    assert all(torch.equal(a, b) for a, b in zip(r0, r1))
    with ops.conv_pair(dev):
        with pytest.raises(RuntimeError, match="already inside a pair"):
            L.check(L.load().codon_conv_pair_begin(), "conv_pair_begin")
    # a held call keeps ITS stream -- two calls issued on different streams are not merged into one grid on
    # pair_end's stream but launched one by one, each where its caller ordered it; a third held-eligible call is refused
    # (it would run ahead of the two held ones) and the first two still leave with pair_end
    side = torch.cuda.Stream(device=dev)
    outs = [ops.new_act(B, 128, H, W, dtype, dev).zero_() for _ in range(2)]
    torch.cuda.synchronize()
    with ops.conv_pair(dev, True) as pr:
        ops.conv2d(Slice(xa, 0, 64), w5a, Slice(outs[0], 64, 64), 5, relu=True)
        with torch.cuda.stream(side):
            ops.conv2d(Slice(xb, 64, 64), w5b, Slice(outs[1], 0, 64), 5, relu=True)
    assert pr.launches == 2
    side.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(outs[0], o0[0]) and torch.equal(outs[1], o0[1])
    outs = [ops.new_act(B, 128, H, W, dtype, dev).zero_() for _ in range(3)]
    with ops.conv_pair(dev, True) as pr:
        ops.conv2d(Slice(xa, 0, 64), w5a, Slice(outs[0], 64, 64), 5, relu=True)
        ops.conv2d(Slice(xb, 64, 64), w5b, Slice(outs[1], 0, 64), 5, relu=True)
        with pytest.raises(RuntimeError, match="third conv call"):
            ops.conv2d(Slice(xa, 0, 64), w5a, Slice(outs[2], 0, 64), 5, relu=True)
    assert pr.launches == 1
    torch.cuda.synchronize()
    assert torch.equal(outs[0], o0[0]) and torch.equal(outs[1], o0[1]) and float(outs[2].float().abs().max()) == 0.0


@pytest.mark.parametrize("hw", [(64, 96), (128, 160)])
def test_conv_pair_fp32_small_grid(hw):
    """fp32: the pair form exists for the small-grid kernels: one launch, same bits -- plain, chained 1x1 and gated + emitting.
    One 64 x 96 image (48 tiles of 4 x 32) pairs the cout-split kernels, one 128 x 160 image (160 tiles) the unsplit ones
    (against cout-split lone launches).  A large grid launches at once (two launches)."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, (H, W) = 1, hw
    q = lambda c, seed: _rand((B, c, H, W), seed).to(dev)
    xa, xb = q(128, 1), q(128, 2)
    wt = lambda co, ci, k, seed, mode=L.PACK_FWD: ops.packed_weight(_rand((co, ci, k, k), seed, (2.0 / (k * k * co)) ** 0.5).to(dev), mode, torch.float32)
    w5a, w5b, w3a, w3b, w5 = wt(64, 64, 5, 3), wt(64, 64, 5, 4), wt(64, 64, 3, 5), wt(64, 64, 3, 6), wt(128, 128, 5, 7)
    w1 = wt(64, 128, 1, 8, L.PACK_CHAIN1X1)
    ch, sp = torch.rand((B, 64), device=dev), torch.rand((B, 1, H, W), device=dev)

    def run(paired):
        o = [torch.zeros((B, 128, H, W), device=dev) for _ in range(5)]
        ns = []
        with ops.conv_pair(dev, paired) as pr:
            ops.conv2d(Slice(xa, 0, 64), w5a, Slice(o[0], 64, 64), 5, relu=True)
            ops.conv2d(Slice(xb, 64, 64), w5b, Slice(o[0], 0, 64), 5, relu=True)
        ns.append(pr.launches)
        with ops.conv_pair(dev, paired) as pr:
            ops.conv2d(Slice(xa, 64, 64), w3a, Slice(o[1], 0, 64), 3, relu=True)
            ops.conv2d(Slice(xb, 0, 64), w3b, Slice(o[1], 64, 64), 3, relu=True)
        ns.append(pr.launches)
        with ops.conv_pair(dev, paired) as pr:
            ops.conv_chain1x1(Slice(xa), w5, w1, Slice(o[2], 64, 64))
            ops.conv_chain1x1(Slice(xb), w5, w1, Slice(o[2], 0, 64))
        ns.append(pr.launches)
        with ops.conv_pair(dev, paired) as pr:
            ops.conv2d_gated(Slice(o[2], 64, 64), Slice(xa, 64, 64), ch, sp, w5a, Slice(o[3], 0, 64), 5, relu=True, emit=Slice(o[4], 64, 64))
            ops.conv2d_gated(Slice(o[2], 0, 64), Slice(xa, 0, 64), ch, sp, w5b, Slice(o[3], 64, 64), 5, relu=True, emit=Slice(o[4], 0, 64))
        ns.append(pr.launches)
        return o, ns

    o0, _ = run(False)
    o1, ns = run(True)
    assert ns == [1, 1, 1, 1], ns
    assert all(torch.equal(a, b) for a, b in zip(o0, o1))
    big_x = _rand((16, 128, 96, 128), 9).to(dev)          # 16 x 12 x 4 = 768 tiles of 8 x 32: priced onto 4 x 32 tiles two per CU
    yb = torch.zeros((16, 128, 96, 128), device=dev)       # (grid_mode), which is not a pair form
    with ops.conv_pair(dev) as pr:
        ops.conv2d(Slice(big_x, 0, 64), w5a, Slice(yb, 0, 64), 5)
        ops.conv2d(Slice(big_x, 64, 64), w5b, Slice(yb, 64, 64), 5)
    assert pr.launches == 0                # both launched at once, nothing was held


def _unsplit_repeat(B, H, W):
    """How often a small input has to be repeated along the batch for its launch to leave the cout-split regime (more than 192
    tiles of 4 x 32) while staying a small grid: the repeated launch runs the UNSPLIT 4 x 32 kernel."""
    nblk4 = B * ((W + 31) // 32) * ((H + 3) // 4)
    return 192 // nblk4 + 1


@pytest.mark.parametrize("shape", [(1, 64, 96), (1, 37, 70), (2, 33, 40), (1, 1, 1), (1, 128, 128)])
def test_fp32_cout_split_chained_conv_is_bit_identical(shape):
    """conv_mfma_f32 CSPLIT (round 5): a small-grid conv5x5 128->128 + chained 1x1 runs as 2 x 32 tiles whose four waves are
    2 rows x 2 cout halves (half the serial MFMA chain per wave; the 1x1's operands meet through LDS).  Every output is the same
    fma chain in the same order: bit-identical to the unsplit small-grid kernel -- which the same images take when they are
    part of a batch of more than 192 tiles -- with and without the residual and the materialised intermediate; alone, and
    as one of a pair (codon_conv_pair_begin / _end: one grid of both launches)."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    rep = _unsplit_repeat(B, H, W)
    x = _rand((B, 128, H, W), 1).to(dev)
    res = _rand((B, 64, H, W), 2).to(dev)
    w5 = ops.packed_weight(_rand((128, 128, 5, 5), 3, (2.0 / (25 * 128)) ** 0.5).to(dev), L.PACK_FWD, torch.float32)
    w1 = ops.packed_weight(_rand((64, 128, 1, 1), 4, 0.1).to(dev), L.PACK_CHAIN1X1, torch.float32)

    def run(xs, rs, use_res, use_mid, pair):
        b = xs.shape[0]
        o = torch.full((b, 128, H, W), float("nan"), device=dev)
        mid = torch.full((b, 128, H, W), float("nan"), device=dev)
        with ops.conv_pair(dev, pair) as pr:
            ops.conv_chain1x1(Slice(xs), w5, w1, Slice(o, 64, 64), mid=Slice(mid) if use_mid else None,
                              residual=Slice(rs) if use_res else None)
            if pair:
                ops.conv_chain1x1(Slice(xs), w5, w1, Slice(o, 0, 64), mid=None, residual=Slice(rs) if use_res else None)
        assert not pair or pr.launches == 1
        return o, mid

    for use_res, use_mid in ((False, False), (True, False), (True, True)):
        o_u, mid_u = run(x.repeat(rep, 1, 1, 1), res.repeat(rep, 1, 1, 1), use_res, use_mid, False)     # unsplit
        o_s, mid_s = run(x, res, use_res, use_mid, False)                                               # cout split, alone
        assert torch.equal(o_u[:B, 64:], o_s[:, 64:]) and torch.isnan(o_s[:, :64]).all()
        if use_mid:
            assert torch.equal(mid_u[:B], mid_s)
        if not use_mid:
            o_p, _ = run(x, res, use_res, False, True)                                                  # cout split, as a pair
            assert torch.equal(o_p[:, 64:], o_s[:, 64:]) and torch.equal(o_p[:, :64], o_s[:, 64:])


@pytest.mark.parametrize("k,cin,cout", [(5, 64, 64), (3, 64, 64), (3, 128, 64), (3, 64, 128), (5, 128, 128)])
def test_fp32_cout_split_plain_conv_is_bit_identical(k, cin, cout):
    """The same split for a small-grid plain conv (the trunk's conv8 / conv9 / conv11 at one image per call, alone; the two
    streams' convs of a block as a pair), every epilogue variant: ReLU, residual, ReLU mask, accumulate, mask of the sum."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    for (B, H, W) in ((1, 64, 96), (1, 37, 70), (1, 2, 3)):
        rep = _unsplit_repeat(B, H, W)
        x = _rand((B, cin, H, W), 1).to(dev)
        r = _rand((B, cout, H, W), 2).to(dev)
        prev = _rand((B, cout, H, W), 3).to(dev)
        wp = ops.packed_weight(_rand((cout, cin, k, k), 4, (2.0 / (k * k * cout)) ** 0.5).to(dev), L.PACK_FWD, torch.float32)
        for kw in (dict(relu=True), dict(residual=r), dict(relu_mask=r), dict(accumulate=True),
                   dict(relu_mask=r, accumulate=True, mask_sum=True)):
            def run(n, pair):
                y = prev.repeat(n, 1, 1, 1)
                y2 = prev.repeat(n, 1, 1, 1)
                kws = {a: (Slice(v.repeat(n, 1, 1, 1)) if isinstance(v, torch.Tensor) else v) for a, v in kw.items()}
                with ops.conv_pair(dev, pair) as pr:
                    ops.conv2d(Slice(x.repeat(n, 1, 1, 1)), wp, Slice(y), k, **kws)
                    if pair:
                        ops.conv2d(Slice(x.repeat(n, 1, 1, 1)), wp, Slice(y2), k, **kws)
                assert not pair or pr.launches == 1
                return y, y2

            y_u, _ = run(rep, False)              # unsplit (more than 192 tiles)
            y_s, _ = run(1, False)                # cout split, alone
            y_p, y_p2 = run(1, True)              # cout split, as a pair
            assert torch.equal(y_u[:B], y_s), (k, cin, cout, H, W, list(kw))
            assert torch.equal(y_p, y_s) and torch.equal(y_p2, y_s), (k, cin, cout, H, W, list(kw))


def test_fp32_grid_modes_are_bit_identical():
    """conv_mfma_f32 grid_mode (round 5): a launch of a few rounds of workgroups picks 8 x 32 tiles, 4 x 32 tiles two per CU or
    4 x 32 tiles one per CU by a cost model of its rounds.  At 300 x 463 one image is priced onto 4 x 32 one-per-CU for the
    chained conv (five of them onto 4 x 32 two-per-CU), at 370 x 463 onto 4 x 32 two-per-CU (five onto 8 x 32): every image of the batch must come
    out with the bits it gets on its own -- plain 5x5 (all epilogues that the one-image forward and backward use), gated +
    emitting, and chained 1x1 with and without residual."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    W = 463
    wt = lambda co, ci, k, seed, mode=L.PACK_FWD: ops.packed_weight(_rand((co, ci, k, k), seed, (2.0 / (k * k * co)) ** 0.5).to(dev), mode, torch.float32)
    w5, w1, w564 = wt(128, 128, 5, 1), wt(64, 128, 1, 2, L.PACK_CHAIN1X1), wt(64, 64, 5, 3)
    import ctypes as C
    tiling = lambda b, h: L.load().codon_conv_tiling_f32(C.byref(L.ConvDesc(b, h, W, 128, 128, 5, 128, 0, 128, 0, 0, 0, 0, L.F32)), 1, 0)
    assert (tiling(1, 300), tiling(1, 370), tiling(5, 300), tiling(5, 370)) == \
        (L.TILING_4X32_SOLO, L.TILING_4X32, L.TILING_4X32, L.TILING_8X32)      # the premise: one image and five take different tilings
    for H in (300, 370):
        B = 5
        x = _rand((B, 128, H, W), 4).to(dev)
        r = _rand((B, 64, H, W), 5).to(dev)
        ch, sp = torch.rand((B, 64), device=dev), torch.rand((B, 1, H, W), device=dev)

        def run(xs, rs, chs, sps):
            b = xs.shape[0]
            o = [torch.zeros((b, 128, H, W), device=dev) for _ in range(4)]
            ops.conv_chain1x1(Slice(xs), w5, w1, Slice(o[0], 64, 64))
            ops.conv_chain1x1(Slice(xs), w5, w1, Slice(o[0], 0, 64), residual=Slice(rs))
            ops.conv2d(Slice(xs, 0, 64), w564, Slice(o[1], 0, 64), 5, relu=True)
            ops.conv2d(Slice(xs, 64, 64), w564, Slice(o[1], 64, 64), 5, residual=Slice(rs))
            ops.conv2d(Slice(xs, 64, 64), w564, Slice(o[1], 64, 64), 5, relu_mask=Slice(rs), accumulate=True, mask_sum=True)
            ops.conv2d_gated(Slice(xs, 0, 64), Slice(xs, 64, 64), chs, sps, w564, Slice(o[2], 0, 64), 5, relu=True, emit=Slice(o[3], 64, 64))
            return o

        full = run(x, r, ch, sp)
        for i in (0, B - 1):
            one = run(x[i:i + 1].contiguous(), r[i:i + 1].contiguous(), ch[i:i + 1].contiguous(), sp[i:i + 1].contiguous())
            for a, b_ in zip(full, one):
                assert torch.equal(a[i:i + 1], b_), (H, i)


@pytest.mark.parametrize("shape", [(1, 128, 128), (1, 37, 70), (1, 64, 96), (2, 5, 3), (1, 1, 1), (1, 100, 200)])
def test_fp32_fused_statistics_against_torch_and_in_every_tiling(shape):
    """Round 6: the fp32 chained conv's statistics epilogue (codon_conv_chain1x1_stats_fwd on fp32 tensors,
    csrc/conv_mfma_f32.hip ST kernels).  (a) against torch on the stored tensor: per-pixel max over the 64 channels and the
    per-strip channel maxima bit for bit, sums to fp32 summation order (F.avg_pool2d / F.max_pool2d / ChannelPool,
    /root/reference/CODON_X4/CAC_module.py:43,47,81); the conv output identical with and without the statistics.
    (b) TILING INVARIANCE -- the reason the partials are per row strip: the image alone (2 x 32 cout-split, 4 x 32 one per CU),
    as a pair launch, and repeated in a batch large enough for 4 x 32 two-per-CU and 8 x 32 tiles gives the SAME BITS in
    pool and partials, so the gates of an image do not depend on the batch it arrives in."""
    import ctypes as C
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    fz = dict(dtype=torch.float32, device=dev)
    x = torch.relu(_rand((B, 128, H, W), 1)).to(dev)
    w5 = _rand((128, 128, 5, 5), 11, (2.0 / (25 * 128)) ** 0.5).to(dev)
    wc = _rand((64, 128, 1, 1), 21, 0.15).to(dev)
    wp, wcp = ops.packed_weight(w5, L.PACK_FWD, torch.float32), ops.packed_weight(wc, L.PACK_CHAIN1X1, torch.float32)
    nt = ops.cac_fused_parts(H, W, torch.float32)
    assert nt == H * ((W + 31) // 32)

    def run(xs, paired=False, choff=64):
        b = xs.shape[0]
        out = torch.full((b, 128, H, W), float("nan"), **fz)
        pool = torch.full((b, 2, H, W), float("nan"), **fz)
        part = torch.full((b, nt, 128, 2), float("nan"), **fz)
        with ops.conv_pair(dev, paired):
            ops.conv_chain1x1(Slice(xs), wp, wcp, Slice(out, 0, 64), stats=(pool, part, choff))
            if paired:
                ops.conv_chain1x1(Slice(xs), wp, wcp, Slice(out, 64, 64), stats=(pool.clone(), part, 64 - choff))
        return out, pool, part

    out, pool, part = run(x)
    ref = torch.zeros((B, 64, H, W), **fz)
    ops.conv_chain1x1(Slice(x), wp, wcp, Slice(ref))
    torch.cuda.synchronize()
    assert torch.equal(out[:, :64], ref) and not torch.isnan(pool).any()
    assert not torch.isnan(part[:, :, 64:]).any() and torch.isnan(part[:, :, :64]).all()     # only its own 64 channels
    assert torch.equal(pool[:, 0], ref.amax(1))
    assert rel_rmse(pool[:, 1].cpu(), ref.double().sum(1).float().cpu()) < 1e-6
    tx = (W + 31) // 32
    padded = torch.full((B, 64, H, tx * 32), float("-inf"), **fz)
    padded[..., :W] = ref
    strips = padded.view(B, 64, H, tx, 32)
    want_max = strips.amax(4).permute(0, 2, 3, 1).reshape(B, nt, 64)
    want_sum = torch.where(torch.isinf(strips), torch.zeros_like(strips), strips).double().sum(4).permute(0, 2, 3, 1).reshape(B, nt, 64)
    assert torch.equal(part[:, :, 64:, 1], want_max)
    assert float((part[:, :, 64:, 0].double() - want_sum).abs().max()) <= 2e-6 * (float(want_sum.abs().max()) + 1e-30)
    # (b) the same image in other launch shapes
    _, pool_p, part_p = run(x, paired=True)
    assert torch.equal(pool_p, pool) and torch.equal(part_p[:, :, 64:], part[:, :, 64:])
    tiling = lambda b: L.load().codon_conv_tiling_f32(C.byref(L.ConvDesc(b, H, W, 128, 128, 5, 128, 0, 128, 0, 0, 0, 0, L.F32)), 1, 0)
    seen = {tiling(B)}
    for rep in (3, 8, 40, 200):
        if B * rep * H * W > 40 * 128 * 128:
            break
        xs = x.repeat(rep, 1, 1, 1)
        seen.add(tiling(B * rep))
        o2, pool2, part2 = run(xs)
        for k in (0, rep - 1):
            sl = slice(k * B, (k + 1) * B)
            assert torch.equal(o2[sl, :64], ref) and torch.equal(pool2[sl], pool), (rep, k)
            assert torch.equal(part2[sl][:, :, 64:], part[:, :, 64:]), (rep, k)
    if H * W >= 64 * 96:
        assert len(seen) >= 3, seen          # the premise: the sizes above really took different tilings


def test_fp32_small_image_bits_do_not_depend_on_the_batch():
    """The fp32 forward of images of at most 32 768 pixels takes its CAC statistics from the conv epilogue (round 6).  One
    128 x 128 image (BASELINE configs[0]) alone, inside a batch of 5 and inside a batch of 32 -- cout-split pairs, 4 x 32 and
    8 x 32 tiles -- comes out with identical bits, and within 1e-4 of the oracle."""
    from codon_amd import CODONNet
    from oracle import codon_oracle as orc
    dev = _dev()
    sd = orc.he_state("x4", seed=61)
    m = CODONNet()
    m.load_state_dict(sd)
    m = m.to(dev).eval()
    g = np.random.default_rng(8)
    x = torch.from_numpy(g.random((32, 1, 128, 128), dtype=np.float32)).to(dev)
    y = torch.from_numpy(g.random((32, 1, 128, 128), dtype=np.float32)).to(dev)
    with torch.no_grad():
        o32 = m(x, y)
        o5 = m(x[3:8].contiguous(), y[3:8].contiguous())
        o1 = m(x[5:6].contiguous(), y[5:6].contiguous())
        ref = orc.forward(sd, x[5:6].cpu(), y[5:6].cpu())
    assert torch.equal(o32[5:6], o1) and torch.equal(o5[2:3], o1)
    assert rmse(o1.cpu(), ref) <= 1e-4 and rel_rmse(o1.cpu(), ref) <= 2e-5


@pytest.mark.parametrize("mode", ["fp16", "bf16"])
def test_16bit_image_bits_do_not_depend_on_the_batch(mode):
    """one image alone (pair launches, the one-launch gate), the same image inside a batch of 5 and inside a
    batch of 32 (lone launches, separate gate launches) give IDENTICAL bits in fp16 / bf16 -- at the reference script's own
    image size (370 x 463, /root/reference/CODON_X4/test.py:116-125)."""
    import ctypes as C
    from codon_amd import CODONNet, _lib as L
    from oracle import codon_oracle as orc
    dev = _dev()
    H, W = 370, 463
    dtype = torch.float16 if mode == "fp16" else torch.bfloat16
    m = CODONNet()
    m.load_state_dict(orc.he_state("x4", seed=62))
    m = m.to(dev).to(dtype).eval()
    g = np.random.default_rng(9)
    x = torch.from_numpy(g.random((32, 1, H, W), dtype=np.float32)).to(dev).to(dtype)
    y = torch.from_numpy(g.random((32, 1, H, W), dtype=np.float32)).to(dev).to(dtype)
    with torch.no_grad():
        o32 = m(x, y)
        o5 = m(x[10:15].contiguous(), y[10:15].contiguous())
        o1 = m(x[12:13].contiguous(), y[12:13].contiguous())
    assert o1.dtype == dtype and bool(torch.isfinite(o1.float()).all())
    assert torch.equal(o32[12:13], o1) and torch.equal(o5[2:3], o1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 128, 128), (1, 37, 70), (2, 33, 40), (1, 1, 1), (1, 370, 463)])
def test_mix53_conv5x5_and_conv3x3_as_one_grid(shape, dtype):
    """Round 6 (VERDICT r5 #1c): conv8 (5x5 64->64) and conv9 (3x3 64->64) of the fusion trunk read the same tensor
    (/root/reference/CODON_X4/CODON_x4.py:123-124).  Held in one pair bracket they leave as ONE grid of two kinds of workgroup
    (mix53) where the launcher has that form -- fp32: the small-grid cout-split kernels; 16-bit: up to the pair limit -- with
    the bits of two separate launches, in either call order; where it has not, as two launches."""
    from codon_amd import _lib as L, ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    if dtype == torch.float32:
        x = torch.relu(_rand((B, 64, H, W), 5)).to(dev)
        new = lambda: torch.full((B, 128, H, W), float("nan"), device=dev)
    else:
        x = ops.from_nchw(torch.relu(_rand((B, 64, H, W), 5)).to(dev), dtype)
        new = lambda: ops.new_act(B, 128, H, W, dtype, dev).fill_(float("nan"))
    wt = lambda k, seed: ops.packed_weight(_rand((64, 64, k, k), seed, (2.0 / (k * k * 64)) ** 0.5).to(dev), L.PACK_FWD, dtype)
    w5, w3 = wt(5, 6), wt(3, 7)

    def run(paired, five_first=True):
        o = new()
        with ops.conv_pair(dev, paired) as pr:
            calls = [lambda: ops.conv2d(Slice(x), w5, Slice(o, 0, 64), 5, relu=True),
                     lambda: ops.conv2d(Slice(x), w3, Slice(o, 64, 64), 3, relu=True)]
            for c in (calls if five_first else calls[::-1]):
                c()
        torch.cuda.synchronize()
        return o, pr.launches

    ref, _ = run(False)
    assert not torch.isnan(ref.float()).any()
    # its own oracle anchor: torch's conv2d on the same (16-bit: rounded) operands (nn.Conv2d(64, 64, k, 1, k // 2, bias=False) +
    # ReLU, /root/reference/CODON_X4/CODON_x4.py:42-43,123-124)
    xr = x if dtype == torch.float32 else ops.to_nchw(x).float()
    rnd = (lambda t: t) if dtype == torch.float32 else (lambda t: t.to(dtype).float())
    r_nchw = ref if dtype == torch.float32 else ops.to_nchw(ref).float()
    for k, seed, sl in ((5, 6, slice(0, 64)), (3, 7, slice(64, 128))):
        w = rnd(_rand((64, 64, k, k), seed, (2.0 / (k * k * 64)) ** 0.5).to(dev))
        want = torch.relu(F.conv2d(xr.double(), w.double(), None, 1, k // 2)).float()
        assert rel_rmse(r_nchw[:, sl].cpu(), want.cpu()) <= (2e-6 if dtype == torch.float32 else _tol(dtype)), k
    for five_first in (True, False):
        got, n = run(True, five_first)
        assert torch.equal(got, ref), (five_first, n)
        import ctypes as C
        if dtype == torch.float32:
            d = L.ConvDesc(B, H, W, 64, 64, 5, 64, 0, 128, 0, 0, 0, L.CONV_RELU, L.F32)
            # fp32: only small-grid launches are held at all -- cout-split ones leave as the mix53 grid, 4 x 32 one-per-CU
            # ones as two launches; larger grids launched at once (pair_end counts the launches IT issued)
            tl = L.load().codon_conv_tiling_f32(C.byref(d), 0, 1)
            expect = {L.TILING_2X32_COUT_SPLIT: 1, L.TILING_4X32_SOLO: 2}.get(tl, 0)
        else:
            expect = 1
        assert n == expect, (n, expect)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(1, 128, 128), (2, 19, 45), (1, 1, 1), (1, 370, 463), (3, 300, 256)])
def test_stem_pair_equals_two_stem_launches(shape, dtype):
    """codon_stem_pair_fwd (round 6): both stems of a forward (/root/reference/CODON_X4/CODON_x4.py:68,71) as one launch --
    the bits of two codon_stem_fwd launches, in every form the lone launch takes (one-pixel form for images of a few thousand
    pixels, four-pixel rows, channel-blocked 16-bit)."""
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = _dev()
    B, H, W = shape
    xa, xb = _rand((B, 1, H, W), 1).to(dev), _rand((B, 1, H, W), 2).to(dev)
    wa, wb = _rand((64, 1, 3, 3), 3, 0.3).to(dev), _rand((64, 1, 3, 3), 4, 0.3).to(dev)
    new = (lambda: torch.full((B, 128, H, W), float("nan"), device=dev)) if dtype == torch.float32 else \
        (lambda: ops.new_act(B, 128, H, W, dtype, dev).fill_(float("nan")))
    ref, got = new(), new()
    ops.stem(xa, wa, Slice(ref, 64, 64))
    ops.stem(xb, wb, Slice(ref, 0, 64))
    ops.stem_pair(xa, wa, Slice(got, 64, 64), xb, wb, Slice(got, 0, 64))
    torch.cuda.synchronize()
    assert not torch.isnan(ref.float()).any() and torch.equal(ref, got)
    # ... and torch's relu(conv2d(1 -> 64, 3x3, pad 1)) on the same operands (CODON_x4.py:24,31,68,71)
    g_nchw = got if dtype == torch.float32 else ops.to_nchw(got).float()
    for xi, wi, sl in ((xa, wa, slice(64, 128)), (xb, wb, slice(0, 64))):
        want = torch.relu(F.conv2d(xi.double(), wi.double(), None, 1, 1)).float()
        assert rel_rmse(g_nchw[:, sl].cpu(), want.cpu()) <= (2e-6 if dtype == torch.float32 else _tol(dtype))
    with pytest.raises(RuntimeError, match="overlap"):
        ops.stem_pair(xa, wa, Slice(got, 0, 64), xb, wb, Slice(got, 0, 64))
