"""Backward parity (row a15): HIP gradients through the C ABI vs (1) autograd gradients recorded from
the imported reference (tests/golden, L1 loss) and (2) the CPU oracle's autograd on fresh inputs.
Tolerance: per-tensor rel-RMSE <= 1e-4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import codon_oracle as orc
from tests.util import BF16_GRAD_CASES, load_case, rel_rmse, rmse, target_for

GRAD_TOL = 1e-4


def _model(variant, sd):
    from codon_amd import CODONNet, CODONNet16
    m = (CODONNet16 if variant == "x16" else CODONNet)()
    m.load_state_dict(sd, strict=True)
    return m.cuda().train()


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(size=shape) * scale).astype(np.float32))


@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 48, 64), (1, 1, 1), (2, 50, 41)])
def test_cac_backward_vs_autograd(shape):
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = torch.device("cuda:0")
    B, H, W = shape
    pre2 = _rand((B, 128, H, W), 1).requires_grad_(True)       # [pre | pre_c]
    in2 = _rand((B, 128, H, W), 2)
    w1, b1 = _rand((8, 128), 3, 0.1).requires_grad_(True), _rand((8,), 4, 0.1).requires_grad_(True)
    w2, b2 = _rand((64, 8), 5, 0.3).requires_grad_(True), _rand((64,), 6, 0.1).requires_grad_(True)
    ws = _rand((1, 2, 5, 5), 7, 0.2).requires_grad_(True)
    g_oc = _rand((B, 128, H, W), 8)
    pre, pre_c = pre2[:, :64], pre2[:, 64:]
    Fcat = torch.cat((pre_c, pre), 1)
    ch = orc.cac_channel(Fcat, w1, b1, w2, b2)
    sp = orc.cac_spatial(Fcat, ws)
    g = ch[:, :, None, None] * sp
    oc = torch.cat((pre * g + in2[:, :64], pre_c * g + in2[:, 64:]), 1)
    oc.backward(g_oc)

    d = lambda t: t.detach().to(dev).contiguous()
    p2 = d(pre2)
    nt = ops.cac_stats_tiles(H, W)
    pooled = torch.empty((B, 2, H, W), device=dev)
    partials = torch.empty((B, nt, 128, 2), device=dev)
    chd = torch.empty((B, 64), device=dev)
    pools = torch.empty((B, 2, 128), device=dev)
    spd = torch.empty((B, 1, H, W), device=dev)
    ops.cac_stats(Slice(p2, 64, 64), Slice(p2, 0, 64), pooled, partials)
    ops.cac_gate(B, H, W, partials, d(w1), d(b1), d(w2), d(b2), chd, pools)
    ops.cac_spatial(pooled, d(ws), spd)
    gocd = d(g_oc)
    g_pre2 = torch.empty((B, 128, H, W), device=dev)
    base = _rand((B, 128, H, W), 9)
    g_in2 = base.to(dev)
    dw1, db1, dw2, db2, dws = ops.cac_backward(
        Slice(gocd, 0, 64), Slice(gocd, 64, 64), Slice(p2, 0, 64), Slice(p2, 64, 64), chd, spd, pooled, pools,
        d(w1), d(b1), d(w2), d(ws), Slice(g_pre2, 0, 64), Slice(g_pre2, 64, 64), Slice(g_in2, 0, 64),
        Slice(g_in2, 64, 64), accumulate_in=True)
    assert rel_rmse(g_pre2.cpu(), pre2.grad) < 1e-5
    assert rel_rmse(g_in2.cpu(), base + g_oc) < 1e-6
    for got, ref, nm in ((dw1, w1.grad, "w1"), (db1, b1.grad, "b1"), (dw2, w2.grad, "w2"), (db2, b2.grad, "b2"),
                         (dws, ws.grad, "ws")):
        assert rel_rmse(got.cpu(), ref) < 2e-5, nm


def test_stencil_backward_pieces():
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = torch.device("cuda:0")
    for (B, H, W) in [(2, 19, 45), (1, 16, 64), (1, 1, 1), (1, 9, 3)]:
        # head: y = conv(t, w_out) + x
        t = torch.relu(_rand((B, 64, H, W), 1)).requires_grad_(True)
        wo = _rand((1, 64, 3, 3), 2, 0.1).requires_grad_(True)
        gy = _rand((B, 1, H, W), 3)
        F.conv2d(t, wo, None, 1, 1).backward(gy)
        g_t = torch.empty((B, 64, H, W), device=dev)
        ops.stencil_1to64(gy.to(dev), wo.detach().to(dev), Slice(g_t), flip=True, mask=Slice(t.detach().to(dev)))
        assert rel_rmse(g_t.cpu(), t.grad * (t.detach() > 0)) < 1e-6
        dwo = torch.empty((1, 64, 3, 3), device=dev)
        ops.conv1ch_wgrad(Slice(t.detach().to(dev)), gy.to(dev), dwo, flip=True)
        assert rel_rmse(dwo.cpu(), wo.grad) < 3e-6
        # stem: s = conv(x, w_in)
        x = _rand((B, 1, H, W), 4)
        wi = _rand((64, 1, 3, 3), 5, 0.3).requires_grad_(True)
        gs = _rand((B, 64, H, W), 6)
        F.conv2d(x, wi, None, 1, 1).backward(gs)
        dwi = torch.empty((64, 1, 3, 3), device=dev)
        ops.conv1ch_wgrad(Slice(gs.to(dev)), x.to(dev), dwi, flip=False)
        assert rel_rmse(dwi.cpu(), wi.grad) < 3e-6


def _relu_mask_flips(model_save, sd, x, y):
    """Number of ReLU'd activations whose mask (value > 0) differs between the HIP forward and the CPU
    oracle forward.  A pre-activation within fp32 noise of zero legitimately lands on either side
    depending on summation order (KAT-0's low-discrepancy weights produce sums of ~1e-9), and the
    gradient is discontinuous there."""
    taps = {}
    with torch.no_grad():
        orc.forward(sd, x, y, taps)
    S = model_save
    pairs = [(S["in2"][:, :64], taps["inputs"]), (S["in2"][:, 64:], taps["inputs_c"]), (S["fuse"], taps["fuse"]),
             (S["t11"], taps["t11"])]
    for i in range(5):
        for k in ("stage", "stage_c", "r2", "r2_c"):
            pairs.append((S[f"blk{i}"][k], taps[f"blk{i}.{k}"]))
    for i in range(3):
        for k in ("stage", "r2"):
            pairs.append((S[f"trunk{i}"][k], taps[f"trunk{i}.{k}"]))
    n = 0
    for h, t in pairs:
        mm = (h.cpu() > 0) != (t > 0)
        if int(mm.sum()):
            # a flip is only legitimate at noise level
            assert float(h.cpu()[mm].abs().max()) < 1e-6 and float(t[mm].abs().max()) < 1e-6
            n += int(mm.sum())
    return n


def _hip_relu_masks(S):
    """The ReLU masks of the HIP forward (its saved activations > 0), in the oracle forward's ReLU call order."""
    on = lambda t: (t.cpu() > 0)
    ms = [on(S["stem"]), on(S["in2"][:, :64]), on(S["stem_c"]), on(S["in2"][:, 64:])]
    for i in range(5):
        b = S[f"blk{i}"]
        ms += [on(b["stage"][:, :64]), on(b["stage_c"][:, 64:]), on(b["stage"][:, 64:]), on(b["stage_c"][:, :64]),
               on(b["r2"]), on(b["r2_c"])]
    ms.append(on(S["fuse"]))
    for i in range(3):
        t = S[f"trunk{i}"]
        ms += [on(t["stage"][:, :64]), on(t["stage"][:, 64:]), on(t["r2"])]
    ms.append(on(S["t11"]))
    return ms


GRAD_CASES = ["kat0_x4_2x32x24", "kat0_x16_2x20x28", "he0_x4_2x24x20_taps", "he2_x16_1x21x27", "he1_x4_2x18x22"]
FLIPS = {}     # case -> number of noise-level ReLU mask flips seen (0 = the strict per-tensor 1e-4 mode ran)


@pytest.mark.parametrize("name", GRAD_CASES)
def test_gradients_match_reference_golden(name):
    from codon_amd.autograd import _CodonFn
    z, variant, sd, x, y = load_case(name)
    m = _model(variant, sd)
    out = m(x.cuda(), y.cuda())
    mask_flips = _relu_mask_flips(out.grad_fn.saved, sd, x, y)
    hip_masks = _hip_relu_masks(out.grad_fn.saved)
    FLIPS[name] = mask_flips
    print(f"[{name}] ReLU mask flips vs the oracle forward: {mask_flips} "
          f"({'strict per-tensor 1e-4' if mask_flips == 0 else f'forced-mask strict mode + conv tensors {2e-3 * mask_flips:.0e}, gate tensors 5e-2, whole vector 1e-4'})")
    tgt = target_for(x)
    loss = (out - tgt.cuda()).abs().mean()
    assert abs(float(loss.detach()) - float(z["loss"])) <= 2e-6 * max(1.0, abs(float(z["loss"])))
    # d(L1)/d(out) = sign(out - tgt)/N is discontinuous: a residual within fp32 noise of zero may flip
    # its sign between two correct implementations.  Feed the REFERENCE's upstream gradient (sign of
    # the golden output's residual) so the test isolates the backward pass, and separately require
    # the HIP path's own signs to agree except where the residual is at noise level.
    ref_out = torch.from_numpy(z["out"])
    g_up = torch.sign(ref_out - tgt) / ref_out.numel()
    flips = (torch.sign(out.detach().cpu() - tgt) != torch.sign(ref_out - tgt))
    assert int(flips.sum()) <= 2 and bool(((ref_out - tgt).abs()[flips] < 1e-5).all())
    out.backward(g_up.cuda())
    # DETERMINISTIC strict mode first, whatever the flips: the oracle's autograd on the HIP forward's own ReLU masks and the
    # same upstream gradient -- every tensor to 1e-4 (the oracle equals the reference's autograd wherever the masks
    # agree: tests/test_oracle.py)
    _, gforced, _ = orc.grads(sd, x, y, tgt, masks=hip_masks, upstream=g_up)
    for k, p in m.named_parameters():
        if k in gforced:
            assert rel_rmse(p.grad.cpu(), gforced[k]) <= GRAD_TOL, (k, "forced masks")
    # ... then against the gradients RECORDED from the reference's autograd
    n = 0
    num = den = worst = 0.0
    for k, p in m.named_parameters():
        if k.startswith("attention_c5") or k.startswith("attention_s5"):
            assert p.grad is None
            continue
        stride = int(z["gradstride." + k])
        got = p.grad.flatten()[::stride].cpu()
        ref = torch.from_numpy(z["grad." + k])
        e = rel_rmse(got, ref)
        nrm = float(z["gradnorm." + k])
        worst = max(worst, e)
        if mask_flips == 0:   # identical ReLU masks: every tensor to 1e-4, norms too
            assert e <= GRAD_TOL, (k, e)
            assert abs(float(p.grad.double().norm()) - nrm) <= 1e-4 * nrm + 1e-12, k
        else:
            # a noise-level mask flip (an activation below 1e-6 on BOTH sides, _relu_mask_flips) legitimately moves the
            # gradients: conv tensors by <= 2e-3 per flip (measured 8.8e-4 with one), the 25 small, cancellation-dominated
            # gate tensors by up to percents (attention_s2.spatial.conv.weight 1.2e-2 with the two flips KAT-0 x4 has since
            # round 6, whose fp32 statistics of small images come out of the conv epilogue in another summation order)
            assert e <= (5e-2 if k.startswith("attention_") else 2e-3 * mask_flips), (k, e, mask_flips)
        num += float((got.double() - ref.double()).pow(2).sum())
        den += float(ref.double().pow(2).sum())
        n += 1
    assert n == 44
    print(f"[{name}] worst per-tensor rel-RMSE {worst:.2e}, whole gradient vector {(num / den) ** 0.5:.2e}")
    assert (num / den) ** 0.5 <= GRAD_TOL     # whole gradient vector, flips or not


def test_golden_gradient_cases_strictness():
    """Per-case BOUND on noise-level ReLU mask flips (each flip individually checked to be < 1e-6 on both sides by
    _relu_mask_flips): how many cases see zero flips depends on summation order and may differ from box to box, so it
    is printed, not asserted -- the strict per-tensor 1e-4 comparison runs for EVERY case on forced masks instead."""
    if set(FLIPS) != set(GRAD_CASES):
        pytest.skip("needs test_gradients_match_reference_golden to have run in this process")
    print("ReLU mask flips per case:", FLIPS)
    assert all(v <= 3 for v in FLIPS.values()), FLIPS


def test_gradients_match_oracle_autograd_random():
    sd = orc.he_state("x4", seed=21)
    g = np.random.default_rng(4)
    B, H, W = 2, 37, 53
    x = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    y = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    tgt = target_for(x)
    loss_ref, gref, out_ref = orc.grads(sd, x, y, tgt)
    m = _model("x4", sd)
    out = m(x.cuda(), y.cuda())
    assert rmse(out.detach().cpu(), out_ref) <= 1e-4
    mask_flips = _relu_mask_flips(out.grad_fn.saved, sd, x, y)
    g_up = (torch.sign(out_ref - tgt) / out_ref.numel()).cuda()   # the oracle's upstream gradient (see above)
    out.backward(g_up)
    num = den = 0.0
    for k, p in m.named_parameters():
        if k in gref:
            e = rel_rmse(p.grad.cpu(), gref[k])
            assert e <= (GRAD_TOL if mask_flips == 0 else 5e-2), (k, e, mask_flips)
            num += float((p.grad.cpu().double() - gref[k].double()).pow(2).sum())
            den += float(gref[k].double().pow(2).sum())
    assert (num / den) ** 0.5 <= GRAD_TOL
    # a second backward accumulates into .grad like any autograd parameter
    out2 = m(x.cuda(), y.cuda())
    out2.backward(g_up)
    assert rel_rmse(m.conv3.weight.grad.cpu(), 2 * gref["conv3.weight"]) <= (GRAD_TOL if mask_flips == 0 else 5e-2)   # as above


def test_gradsync_flat_views_accumulate_in_place():
    """Autograd must accumulate into the flat-buffer views GradSync installs (no re-allocation of .grad),
    and an Adam step through those views must change the parameters."""
    from codon_amd.dist import GradSync
    sd = orc.he_state("x4", seed=5)
    m = _model("x4", sd)
    gs = GradSync(m)
    x, y = orc.kat_inputs(2, 16, 24)
    tgt = target_for(x).cuda()
    opt = torch.optim.Adam(gs.params, lr=1e-3)
    gs.zero_grad()
    (m(x.cuda(), y.cuda()) - tgt).abs().mean().backward()
    assert m.conv3.weight.grad._base is gs.flat and float(gs.flat.abs().sum()) > 0
    n1 = float(gs.flat.norm())
    (m(x.cuda(), y.cuda()) - tgt).abs().mean().backward()         # second backward accumulates
    assert abs(float(gs.flat.norm()) - 2 * n1) < 1e-4 * n1
    w0 = m.conv3.weight.detach().clone()
    opt.step()
    assert not torch.equal(w0, m.conv3.weight)
    # packed-weight cache notices the in-place update (version counter) -> next forward uses new weights
    o1 = m(x.cuda(), y.cuda())
    m2 = _model("x4", {k: v.detach().cpu() for k, v in m.state_dict().items()})
    with torch.no_grad():
        o2 = m2(x.cuda(), y.cuda())
    assert torch.equal(o1.detach(), o2)


def test_bf16_training_gradients_vs_fp32_oracle():
    """bf16 activations/gradients, fp32 accumulate, fp32 master weights and parameter gradients
    (BASELINE configs[2]).  Tolerance: whole gradient vector rel-RMSE <= 5e-2 vs the fp32 oracle's autograd
    (bf16 has 8 mantissa bits: ~4e-3 per rounding, ~40 layers deep); big tensors individually <= 8e-2."""
    sd = orc.he_state("x4", seed=31)
    g = np.random.default_rng(8)
    B, H, W = 2, 40, 48
    x = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    y = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    tgt = target_for(x)
    _, gref, out_ref = orc.grads(sd, x, y, tgt)
    m = _model("x4", sd).set_compute_dtype(torch.bfloat16)
    out = m(x.cuda(), y.cuda())
    assert out.dtype == torch.float32 and rel_rmse(out.detach().cpu(), out_ref) <= 3e-2
    out.backward((torch.sign(out_ref - tgt) / out_ref.numel()).cuda())
    num = den = 0.0
    for k, p in m.named_parameters():
        if k in gref:
            assert p.grad.dtype == torch.float32 and torch.isfinite(p.grad).all(), k
            if gref[k].numel() >= 36864:
                assert rel_rmse(p.grad.cpu(), gref[k]) <= 8e-2, k
            num += float((p.grad.cpu().double() - gref[k].double()).pow(2).sum())
            den += float(gref[k].double().pow(2).sum())
    assert (num / den) ** 0.5 <= 5e-2


BF16_GRAD_RATIO = 1.25      # the rule test_forward_bf16_vs_reference_module_run_in_bf16 uses for the forward
BF16_NOISE_FLOOR = 0.10     # pooled reference-bf16 error at which a tensor's bf16 gradient has < 1 significant digit
BF16_NOISY_RATIO = 1.6      # such a tensor: HIP absolute error <= 1.6 x the reference's own absolute bf16 error (round 6; measured 0.86 ... 1.38)


@pytest.mark.parametrize("name", BF16_GRAD_CASES)
def test_bf16_gradients_vs_reference_bf16_autograd(name):
    """The bf16 BACKWARD pinned to the reference's own bf16 behaviour (tools/make_golden_r4.py): the reference module cast
    to bfloat16, forward + autograd on CPU, against its float64 twin from the same upstream gradient, on 8 input variants.
    Per tensor -- all 44, the 25 CAC parameter tensors (MLP weights / biases, 5x5 spatial conv: CAC_module.py:30-35,88)
    included; see BF16_NOISE_FLOOR below for the one documented exception class -- the HIP bf16 path (bf16 activations / activation gradients, fp32 accumulate) must be no further from fp64
    than BF16_GRAD_RATIO x the reference's bf16 run, both errors pooled (RMS) over the variants: ONE realisation of a
    small tensor's bf16 error is noise (the reference's own spans 1.2e-2 ... 2.8e-1 for attention_s0 over the 8 variants;
    a single-variant ratio test was tried first and measured exactly that), the pooled figure is a statistic.
    Both modes: fp32 master weights + bf16 compute (BASELINE configs[2]) and the whole-module .bfloat16() cast."""
    from tests.util import bf16grad_inputs
    z, variant, sd, _, _ = load_case(name)
    B, H, W = (int(v) for v in z["shape"])
    nv = int(z["nv"])
    for mode in ("fp32 master weights + bf16 compute (configs[2])", "whole-module .bfloat16()"):
        m = _model(variant, sd)
        m = m.set_compute_dtype(torch.bfloat16) if mode.startswith("fp32") else m.bfloat16()
        hip2, ref2, habs, rabs, num_h, num_r, den = {}, {}, {}, {}, 0.0, 0.0, 0.0
        for v in range(nv):
            x, y = bf16grad_inputs(v, B, H, W)
            up = torch.from_numpy(z[f"v{v}.upstream"]).cuda()
            m.zero_grad(set_to_none=True)
            out = m(x.cuda(), y.cuda()) if mode.startswith("fp32") else m(x.cuda().bfloat16(), y.cuda().bfloat16())
            if v == 0:
                assert rel_rmse(out.detach().float().cpu(), z["out_fp64"]) <= 1.25 * rel_rmse(z["out_bf16"], z["out_fp64"])
            out.backward(up.to(out.dtype))
            n = 0
            for k, p in m.named_parameters():
                if k.startswith("attention_c5") or k.startswith("attention_s5"):
                    assert p.grad is None
                    continue
                s = int(z["stride." + k])
                g64 = z[f"v{v}.g64.{k}"].astype(np.float64)
                got = p.grad.detach().float().flatten()[::s].cpu().double().numpy()
                assert np.isfinite(got).all(), k
                d2, n2 = float(((got - g64) ** 2).sum()), float((g64 ** 2).sum())
                e_ref = float(z[f"v{v}.err_sub.{k}"])
                hip2.setdefault(k, []).append(d2 / n2)
                ref2.setdefault(k, []).append(e_ref ** 2)
                habs[k] = habs.get(k, 0.0) + d2                        # squared ABSOLUTE errors, summed over the variants
                rabs[k] = rabs.get(k, 0.0) + e_ref ** 2 * n2
                num_h += d2
                num_r += e_ref ** 2 * n2
                den += n2
                n += 1
            assert n == 44
        ratios = {k: (np.mean(hip2[k]) / np.mean(ref2[k])) ** 0.5 for k in hip2}
        worst = max(ratios.items(), key=lambda kv: kv[1])
        cac = {k: v for k, v in ratios.items() if k.startswith("attention_")}
        conv = {k: v for k, v in ratios.items() if not k.startswith("attention_")}
        print(f"[{name}] {mode}: HIP-bf16 error / reference-bf16 error (both vs fp64, RMS over {nv} variants): worst "
              f"{worst[0]} {worst[1]:.2f}; 19 conv tensors {min(conv.values()):.2f}..{max(conv.values()):.2f} (median "
              f"{float(np.median(list(conv.values()))):.2f}); 25 CAC tensors {min(cac.values()):.2f}..{max(cac.values()):.2f} "
              f"(median {float(np.median(list(cac.values()))):.2f}); whole vector HIP {(num_h / den) ** 0.5:.3e} reference "
              f"{(num_r / den) ** 0.5:.3e}")
        # A tensor whose gradient the REFERENCE's own bf16 run gets wrong by >= 10 % (pooled) carries less than one
        # significant digit in bf16 whoever computes it (measured: the 5x5 spatial-gate weight of block 0 on the 2 x 24 x 20
        # case -- reference 1.2e-2 ... 2.8e-1 over the variants, pooled 0.13; the EXACT fp32 HIP path is already 7e-3 from
        # fp64 there, a cancellation of ~1e5).  The pooled RELATIVE error of such a tensor is set by the one variant whose
        # true gradient happens to cancel furthest (1 / |g64|^2 weights), i.e. by one noise sample of each side; its
        # ABSOLUTE error is a statistic over every variant and element.  Round 6: those tensors -- at most two per case --
        # are held to BF16_NOISY_RATIO x the reference's own error in that absolute measure (both sides' squared errors
        # vs fp64 summed over the variants; measured 0.86 / 1.18 and 1.14 / 1.38 in the two modes); no absolute bound is left.
        noisy = {k for k in ratios if np.mean(ref2[k]) ** 0.5 >= BF16_NOISE_FLOOR}
        if noisy:
            print(f"[{name}] {mode}: bf16-noise-dominated in the reference itself (pooled reference error >= "
                  f"{BF16_NOISE_FLOOR}): " + ", ".join(
                      f"{k} relative: ref {np.mean(ref2[k]) ** 0.5:.2f} HIP {np.mean(hip2[k]) ** 0.5:.2f}; absolute-error ratio "
                      f"HIP / ref {(habs[k] / rabs[k]) ** 0.5:.2f}" for k in sorted(noisy)))
        assert len(noisy) <= 2 and all(k.startswith("attention_s") for k in noisy), noisy
        for k in noisy:
            assert (habs[k] / rabs[k]) ** 0.5 <= BF16_NOISY_RATIO, (mode, k, (habs[k] / rabs[k]) ** 0.5)
        bad = {k: round(float(v), 3) for k, v in ratios.items() if k not in noisy and not v <= BF16_GRAD_RATIO}
        assert not bad, (mode, bad)
        assert len(cac) == 25 and len(conv) == 19
        assert (num_h / den) ** 0.5 <= 1.1 * (num_r / den) ** 0.5, mode


def test_input_gradients_match_oracle_autograd():
    """dL/dx and dL/dy (the reference's autograd provides them when the inputs require grad): the stems' 64 -> 1 dgrad
    through the head stencil, plus the identity path of the final residual add for x."""
    sd = orc.he_state("x4", seed=33)
    g = np.random.default_rng(9)
    B, H, W = 2, 21, 38
    x = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    y = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    up = torch.from_numpy(g.standard_normal(size=(B, 1, H, W)).astype(np.float32)) / (B * H * W)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    out_ref = orc.forward(sd, xr, yr)
    out_ref.backward(up)
    m = _model("x4", sd)
    xd, yd = x.cuda().requires_grad_(True), y.cuda().requires_grad_(True)
    out = m(xd, yd)
    flips = _relu_mask_flips(out.grad_fn.saved, sd, x, y)
    out.backward(up.cuda())
    tol = GRAD_TOL if flips == 0 else 5e-2
    assert rel_rmse(xd.grad.cpu(), xr.grad) <= tol, flips
    assert rel_rmse(yd.grad.cpu(), yr.grad) <= tol, flips
    # only one input asks: the other stays None
    xd2 = x.cuda().requires_grad_(True)
    m(xd2, y.cuda()).backward(up.cuda())
    assert rel_rmse(xd2.grad.cpu(), xr.grad) <= tol


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_training_steps_reduce_the_loss(dtype):
    """End-to-end sanity of the whole training path (forward, L1+SSIM loss kernels, backward kernels, GradSync flat
    buffer, Adam): a few steps on one fixed batch must bring the loss down substantially."""
    from codon_amd import CODONNet
    from codon_amd.dist import GradSync
    from codon_amd.metrics import L1SSIMLoss
    torch.manual_seed(3)
    m = CODONNet().cuda().train()
    if dtype is not None:
        m.set_compute_dtype(dtype)
    gs = GradSync(m)
    opt = torch.optim.Adam(gs.params, lr=2e-4)
    crit = L1SSIMLoss(1.0, 1.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand((2, 1, 48, 64), generator=g, device="cuda")
    y = torch.rand((2, 1, 48, 64), generator=g, device="cuda")
    tgt = (0.5 * x + 0.25).clamp(0, 1)
    losses = []
    for _ in range(25):
        gs.zero_grad()
        loss = crit(m(x, y).float(), tgt)
        loss.backward()
        gs.all_reduce_grads()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.5 * losses[0], losses


@pytest.mark.parametrize("dtype", [None, torch.bfloat16])
def test_recompute_switch_gives_identical_gradients(dtype):
    """model.set_recompute(): the 13 `stage` tensors are not kept, the backward re-runs the sibling convs from the saved
    block inputs -- same kernels on the same operands, so every gradient is bit-identical and less memory is held."""
    sd = orc.he_state("x4", seed=41)
    g = np.random.default_rng(12)
    B, H, W = 2, 40, 56
    x = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32)).cuda()
    y = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32)).cuda()
    up = torch.from_numpy(g.standard_normal(size=(B, 1, H, W)).astype(np.float32)).cuda() / (B * H * W)
    grads, peaks = [], []
    for rec in (False, True):
        m = _model("x4", sd)
        if dtype is not None:
            m.set_compute_dtype(dtype)
        m.set_recompute(rec)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        out = m(x, y)
        assert (out.grad_fn.saved["blk0"]["stage"] is None) == rec
        out.backward(up)
        torch.cuda.synchronize()
        peaks.append(torch.cuda.max_memory_allocated())
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        del m, out
    assert len(grads[0]) == 44 and all(torch.equal(grads[0][k], grads[1][k]) for k in grads[0])
    assert peaks[1] < peaks[0]


def test_packed_weight_verification_catches_data_writes(monkeypatch):
    """A write through `.data` after the first forward does not bump Tensor._version, so the packed-weight cache cannot
    see it.  DEFAULT behaviour (no environment switch): every forward carries one weight-checksum launch
    (model._WeightGuard, csrc/wsum.hip); the forward enqueued right after the write still runs on the stale packed
    weights, the NEXT call -- or check_packed() at any synchronisation point -- raises.  CODON_VERIFY_PACKED=1 (debug)
    raises in the same call.  model.invalidate_packed() is the remedy."""
    import codon_amd.model as M
    assert M.WEIGHT_GUARD and not M.VERIFY_PACKED
    sd = orc.he_state("x4", seed=43)
    m = _model("x4", sd).eval()
    x = torch.rand((1, 1, 16, 24), device="cuda")
    with torch.no_grad():
        o0 = m(x, x)
        assert torch.equal(m(x, x), o0)
        m.check_packed()                         # clean so far
        m.conv3.weight.data.mul_(0.5)            # the reference's own init idiom writes through .data (CODON_x4.py:50-53)
        o1 = m(x, x)                             # enqueued on stale packed weights; its checksum launch notices
        torch.cuda.synchronize()
        assert torch.equal(o1, o0)
        with pytest.raises(RuntimeError, match="stale"):
            m(x, x)
        with pytest.raises(RuntimeError, match="stale"):
            m.check_packed()
        m.invalidate_packed()
        o2 = m(x, x)
        assert not torch.equal(o2, o0)
        m.check_packed()
        # a single changed BIT in the last weight of the list is seen too, and in training mode as well
        m.conv11.weight.data.view(torch.int32)[-1, -1, -1, -1] ^= 1
        m(x, x)
        with pytest.raises(RuntimeError, match="stale"):
            m.check_packed()
        m.invalidate_packed()
        # visible updates (optimizer-style in-place ops bump _version) never trip it
        m.conv3.weight.mul_(2.0)
        o3 = m(x, x)
        m.check_packed()
        assert not torch.equal(o3, o2)
        # the remedy applied WITHOUT a synchronisation in between: the stale forward's checksum launch may still be in
        # flight when invalidate_packed() runs -- its late store goes to the retired flag word, not the new one
        m.conv3.weight.data.mul_(0.5)
        m(x, x)
        m.invalidate_packed()
        for _ in range(3):
            m(x, x)
        m.check_packed()
        # a trip nobody has been told about is not lost when the guard is reset -- .half() / .to() /
        # load_state_dict() / invalidate_packed() right after the stale forward report it (RuntimeWarning) before they reset
        for remedy in (lambda: m.load_state_dict(m.state_dict()), lambda: m.to("cuda:0", torch.float32).float(),
                       lambda: m.invalidate_packed()):
            m(x, x)
            m.check_packed()
            m.conv3.weight.data.mul_(0.5)
            m(x, x)                              # stale; no synchronisation, no further forward
            with pytest.warns(RuntimeWarning, match="stale packed weights"):
                remedy()
            m(x, x)
            m.check_packed()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("error")       # and a clean model resets silently
            m.load_state_dict(m.state_dict())
            m.invalidate_packed()
        m(x, x)                                  # packed images of the current weights exist again
        m.check_packed()
        # the debug switch: same-call detection
        m.conv3.weight.data.mul_(0.5)
        monkeypatch.setattr(M, "VERIFY_PACKED", True)
        with pytest.raises(RuntimeError, match="stale"):
            m(x, x)
        monkeypatch.setattr(M, "VERIFY_PACKED", False)
        m.invalidate_packed()
        m(x, x)
        m.check_packed()


def test_weight_guard_in_graph_replay_and_threads():
    """The captured forward carries a COMPARING checksum launch (the reference comes from GraphedCODON's warm-up runs):
    a replay after a `.data` write trips the guard; a second host thread on its own stream has its own guard state."""
    import threading
    from codon_amd.graph import GraphedCODON
    sd = orc.he_state("x4", seed=44)
    m = _model("x4", sd).eval()
    x = torch.rand((1, 1, 24, 32), device="cuda")
    with torch.no_grad():
        gm = GraphedCODON(m, x, x)
        o0 = gm(x, x)
        assert not gm.stale(synchronize=True) and torch.equal(gm(x, x), m(x, x))
        errs = []

        def other():
            try:
                with torch.cuda.stream(torch.cuda.Stream()):
                    for _ in range(3):
                        assert torch.equal(m(x, x), o0)
                    torch.cuda.current_stream().synchronize()
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        t = threading.Thread(target=other)
        t.start()
        for _ in range(3):
            m(x, x)
        t.join()
        assert not errs, errs
        m.check_packed()
        m.conv8.weight.data.add_(0.01)
        gm(x, x)                                 # replay on stale packed weights: its checksum launch notices
        assert gm.stale(synchronize=True)
        with pytest.raises(RuntimeError):
            gm(x, x)


@pytest.mark.gpu
def test_full_size_training_step_is_deterministic():
    """16 x 480 x 640 bf16, forward + backward three times on the same inputs: output and every parameter gradient are
    bit-identical between runs.  All reductions are fixed-order; what this guards is the hand-synchronised code -- LDS-DMA
    tiles read through inline-asm transposing reads with counted `s_waitcnt lgkmcnt`, the barrier placed ahead of a tile's
    last k-step (conv_wgrad_c8.hip), the emitting gated convs, the fused 1x1 backward: a race there is a mismatch here."""
    import codon_amd
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = codon_amd.CODONNet().to(dev)
    net.set_compute_dtype(torch.bfloat16)
    net.train()
    B, H, W = 16, 480, 640
    x, y = torch.rand((B, 1, H, W), device=dev), torch.rand((B, 1, H, W), device=dev)
    gy = torch.randn((B, 1, H, W), device=dev)
    ref = None
    for _ in range(3):
        net.zero_grad(set_to_none=True)
        out = net(x, y)
        out.backward(gy)
        g = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        g["__out__"] = out.detach().clone()
        if ref is None:
            ref = g
            assert all(torch.isfinite(v).all() for v in g.values())
        else:
            assert [n for n in ref if not torch.equal(ref[n], g[n])] == []
