"""End-to-end parity of codon_amd.CODONNet (HIP path through the C ABI) against
 (1) the golden fixtures recorded from the imported reference (tests/golden), and
 (2) the CPU oracle on fresh seeded inputs.
Tolerance: RMSE <= 1e-4 absolute on the network output (north_star), with the fp32 noise
floor of the reference itself at 1.2e-5 (He-init, output std ~5)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import codon_oracle as orc
from tests.util import BF16_REF_CASES, FP16_REF_CASES, GOLDEN_CASES, load_case, rel_rmse, rmse

RMSE_TOL = 1e-4


def _model(variant, sd):
    from codon_amd import CODONNet, CODONNet16
    m = (CODONNet16 if variant == "x16" else CODONNet)()
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_forward_matches_golden(name):
    z, variant, sd, x, y = load_case(name)
    m = _model(variant, sd)
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
    assert o.shape == x.shape and o.dtype == torch.float32 and o.is_cuda
    e = rmse(o.cpu(), z["out"])
    assert e <= RMSE_TOL, e
    assert rmse(o.cpu(), z["out_fp64"]) <= RMSE_TOL
    # well inside the bar: same order as the reference's own fp32 noise
    assert rel_rmse(o.cpu(), z["out_fp64"]) <= 2e-5


def test_intermediates_match_golden():
    z, variant, sd, x, y = load_case("kat0_x4_1x13x11_taps")
    m = _model(variant, sd)
    save = {}
    with torch.no_grad():
        m._forward_impl(x.cuda(), y.cuda(), save)
    for i in range(5):
        blk = save[f"blk{i}"]
        assert rel_rmse(blk["pre2"][:, :64].cpu(), z[f"tap.blk{i}.pre"]) < 1e-5
        assert rel_rmse(blk["pre2"][:, 64:].cpu(), z[f"tap.blk{i}.pre_c"]) < 1e-5
        assert rel_rmse(blk["ch"].cpu(), z[f"tap.blk{i}.ch"]) < 1e-5
        assert rel_rmse(blk["sp"].cpu(), z[f"tap.blk{i}.sp"]) < 1e-5
    assert rel_rmse(save["oc"][:, :64].cpu(), z["tap.blk4.out"]) < 1e-5
    assert rel_rmse(save["oc"][:, 64:].cpu(), z["tap.blk4.out_c"]) < 1e-5
    assert rel_rmse(save["fuse"].cpu(), np.maximum(z["tap.fuse.prerelu"], 0)) < 1e-5


@pytest.mark.parametrize("shape,seed", [((2, 40, 72), 0), ((1, 75, 93), 1), ((3, 8, 32), 2), ((1, 128, 128), 3)])
def test_forward_matches_oracle_random(shape, seed):
    """Config-1 size (1x128x128) and ragged sizes; He-init weights, uniform [0,1] inputs."""
    B, H, W = shape
    sd = orc.he_state("x4", seed=10 + seed)
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    y = torch.from_numpy((g.integers(0, 256, size=(B, 1, H, W)) / 255.0).astype(np.float32))
    with torch.no_grad():
        ref = orc.forward(sd, x, y)
    m = _model("x4", sd)
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
    assert rmse(o.cpu(), ref) <= RMSE_TOL
    assert rel_rmse(o.cpu(), ref) <= 2e-5


@pytest.mark.parametrize("shape", [(1, 370, 463), (1, 375, 450), (1, 247, 343)])
def test_forward_at_the_reference_scripts_image_sizes(shape):
    """The images the reference script actually feeds (Middlebury crops under CODON_X*/input_depth: odd widths, one
    image per call, test.py:116-125): fp32 vs the oracle at the 1e-4 bar, and the script's own precision (.half())
    against the same oracle at the fp16 tolerance."""
    B, H, W = shape
    sd = orc.he_state("x4", seed=70 + H)
    g = np.random.default_rng(H)
    x = torch.from_numpy(g.random((B, 1, H, W), dtype=np.float32))
    y = torch.from_numpy((g.integers(0, 256, size=(B, 1, H, W)) / 255.0).astype(np.float32))
    with torch.no_grad():
        ref = orc.forward(sd, x, y)
    m = _model("x4", sd)
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
    assert rmse(o.cpu(), ref) <= RMSE_TOL and rel_rmse(o.cpu(), ref) <= 2e-5
    # the script's precision at these sizes, against the reference module's OWN .half() run on these very inputs
    # (tests/golden/fp16ref_script_sizes.npz, tools/make_golden_r2.py: every 7th pixel of its fp16 and fp64 outputs, recorded
    # in the build container -- the GPU box's CPU runs fp16 convs far too slowly to recompute it there): no further from fp64
    # than 1.25 x the reference.  (The same ratio on a full recorded image: test_forward_half_vs_reference_module_run_in_half.)
    import os
    from tests.util import GOLD
    z = np.load(os.path.join(GOLD, "fp16ref_script_sizes.npz"))
    tag, sub = f"{H}x{W}", int(z["sub"])
    assert float(z[tag + ".x00"]) == float(x[0, 0, 0, 0]) and float(z[tag + ".y_last"]) == float(y[0, 0, -1, -1])
    assert rel_rmse(ref.reshape(-1)[::sub], z[tag + ".out_fp64_sub"]) <= 2e-5          # same weights, same inputs
    ref_err = rel_rmse(z[tag + ".out_fp16_sub"].astype(np.float32), z[tag + ".out_fp64_sub"])
    mh = _model("x4", sd).half()
    with torch.no_grad():
        oh = mh(x.cuda().half(), y.cuda().half())
    assert oh.dtype == torch.float16
    our_err = rel_rmse(oh.float().cpu().reshape(-1)[::sub], z[tag + ".out_fp64_sub"])
    print(f"{shape}: HIP fp16 vs fp64 {our_err:.3e}, reference .half() {ref_err:.3e}, ratio {our_err / ref_err:.3f}")
    assert abs(ref_err - float(z[tag + ".ref_err_full"])) <= 0.05 * ref_err          # the subsample represents the image
    assert our_err <= 1.25 * ref_err, (our_err, ref_err)


def test_batch_independence_and_determinism():
    """Images are independent units (no op mixes samples): a batch equals its images
    run one by one, bit for bit; and two runs are bit-identical (fixed-order reductions)."""
    sd = orc.he_state("x4", seed=3)
    g = np.random.default_rng(5)
    x = torch.from_numpy(g.uniform(0, 1, size=(3, 1, 24, 40)).astype(np.float32)).cuda()
    y = torch.from_numpy(g.uniform(0, 1, size=(3, 1, 24, 40)).astype(np.float32)).cuda()
    m = _model("x4", sd)
    with torch.no_grad():
        o1, o2 = m(x, y), m(x, y)
        singles = torch.cat([m(x[i:i + 1], y[i:i + 1]) for i in range(3)])
    assert torch.equal(o1, o2)
    assert torch.equal(o1, singles)


def test_full_size_properties():
    """BASELINE config-2 spatial size (480x640) at batch 2: the oracle is too slow here, so check
    size-independent properties: (a) zero trunk/head weights => output == input depth exactly
    (global residual, CODON_x4.py:131); (b) translation of a tile interior: a crop run alone equals
    the same region of the full run away from the 40-pixel receptive-field border is NOT expected
    (global pools) -- instead check batch independence at full size."""
    sd = orc.kat_state("x4")
    m = _model("x4", sd)
    g = np.random.default_rng(0)
    x = torch.from_numpy(g.uniform(0, 1, size=(2, 1, 480, 640)).astype(np.float32)).cuda()
    y = torch.from_numpy(g.uniform(0, 1, size=(2, 1, 480, 640)).astype(np.float32)).cuda()
    with torch.no_grad():
        o = m(x, y)
        o0 = m(x[:1].contiguous(), y[:1].contiguous())
    assert torch.isfinite(o).all()
    assert torch.equal(o[:1], o0)
    sd0 = dict(sd)
    sd0["output.weight"] = torch.zeros_like(sd["output.weight"])
    m0 = _model("x4", sd0)
    with torch.no_grad():
        assert torch.equal(m0(x, y), x)


# ---- bf16 compute (fp32 master weights): rel-RMSE <= 3e-2 vs the fp32 oracle (the reference's
# own bf16 CPU run sits at 1.8e-2) -------------------------------------------------------------------
@pytest.mark.parametrize("name", ["he0_x4_2x24x20_taps", "kat0_x4_2x32x24", "he0_x16_1x33x9"])
def test_forward_bf16_matches_golden(name):
    z, variant, sd, x, y = load_case(name)
    m = _model(variant, sd).set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
    assert o.dtype == torch.float32
    assert rel_rmse(o.cpu(), z["out_fp64"]) <= 3e-2
    # .bfloat16() modules + bf16 inputs (the reference-style whole-module cast) take the same path
    mb = _model(variant, sd).bfloat16()
    with torch.no_grad():
        ob = mb(x.cuda().bfloat16(), y.cuda().bfloat16())
    assert ob.dtype == torch.bfloat16
    assert rel_rmse(ob.float().cpu(), z["out_fp64"]) <= 4e-2


@pytest.mark.parametrize("name", BF16_REF_CASES)
def test_forward_bf16_vs_reference_module_run_in_bf16(name):
    """The reference nn.Module cast to bfloat16 and run on CPU (tools/make_golden_r2.py) is itself 0.9-1.2e-2 away from
    its fp64 run.  The HIP bf16 path (bf16 storage, fp32 accumulate) must be (a) no further from fp64 than 1.25x the
    reference's own bf16 error, and (b) within 3e-2 of the reference's bf16 output."""
    z, variant, sd, x, y = load_case(name)
    m = _model(variant, sd).bfloat16()
    with torch.no_grad():
        o = m(x.cuda().bfloat16(), y.cuda().bfloat16()).float().cpu()
    ref_err = rel_rmse(z["out_bf16"], z["out_fp64"])
    our_err = rel_rmse(o, z["out_fp64"])
    assert 5e-3 < ref_err < 3e-2
    assert our_err <= 1.25 * ref_err, (our_err, ref_err)
    assert rel_rmse(o, z["out_bf16"]) <= 3e-2
    # fp32 master weights + bf16 compute (configs[2]'s mode) is at least as close
    m2 = _model(variant, sd).set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        o2 = m2(x.cuda(), y.cuda()).cpu()
    assert rel_rmse(o2, z["out_fp64"]) <= 1.25 * ref_err


def test_forward_bf16_random_128():
    sd = orc.he_state("x4", seed=13)
    g = np.random.default_rng(3)
    x = torch.from_numpy(g.uniform(0, 1, size=(1, 1, 128, 128)).astype(np.float32))
    y = torch.from_numpy((g.integers(0, 256, size=(1, 1, 128, 128)) / 255.0).astype(np.float32))
    with torch.no_grad():
        ref = orc.forward(sd, x, y)
    m = _model("x4", sd).set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
    assert rel_rmse(o.cpu(), ref) <= 3e-2


# ---- fp16: the reference script's own inference precision (model.cuda().half(), test.py:52,122-123) -------
@pytest.mark.parametrize("name", ["he0_x4_2x24x20_taps", "kat0_x4_2x32x24", "he0_x16_1x33x9", "kat0_x4_1x1x1"])
def test_forward_half_like_reference_script(name):
    z, variant, sd, x, y = load_case(name)
    m = _model(variant, sd).half()                      # exactly what test.py does
    with torch.no_grad():
        o = m(x.cuda().half(), y.cuda().half())
    assert o.dtype == torch.float16 and o.shape == x.shape
    # fp16 has 11 significand bits: ~8x tighter than bf16.  Bound: 1.25 x the error of the reference module's own .half()
    # run on the same case (tests/golden/fp16ref_*.npz).  One pixel (kat0_x4_1x1x1: 0.000516, fp16 spacing 4.8e-7 = 9e-4
    # relative) is one noise sample per side, not a statistic: there the bound is the reference's error plus two spacings.
    zr, _, _, _, _ = load_case("fp16ref_" + name.replace("_taps", ""))
    assert np.array_equal(zr["out_fp64"], z["out_fp64"])
    ref_err = rel_rmse(zr["out_fp16"].astype(np.float32), z["out_fp64"])
    our_err = rel_rmse(o.float().cpu(), z["out_fp64"])
    print(f"{name}: HIP fp16 vs fp64 {our_err:.3e}, reference .half() {ref_err:.3e}")
    if x.numel() == 1:
        assert our_err <= ref_err + 2 * 2.0 ** -11, (our_err, ref_err)
    else:
        assert our_err <= 1.25 * ref_err, (our_err, ref_err)
    # test.py:66,125 calls model(...) in eval mode WITHOUT torch.no_grad(): served by the inference schedule,
    # detached, bit-identical to the no_grad call; in train mode fp16 is refused, not silently wrong
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        og = m(x.cuda().half(), y.cuda().half())
    assert not og.requires_grad and torch.equal(og, o)
    m.train()
    with pytest.raises(NotImplementedError):
        m(x.cuda().half(), y.cuda().half())


@pytest.mark.parametrize("name", FP16_REF_CASES)
def test_forward_half_vs_reference_module_run_in_half(name):
    """Round 6: the reference nn.Module after `.half()` on `.half()` inputs, run on CPU (tools/make_golden_r2.py) -- the
    only precision /root/reference/CODON_X4/test.py:52,122-125 ever runs -- is itself 1.0e-3 ... 1.44e-3 from its fp64
    run.  The HIP fp16 path (fp16 storage, fp32 accumulate) must be (a) no further from fp64 than 1.25 x the reference's
    OWN fp16 error on the same case and (b) within 2.5 x that of the reference's fp16 output (two independent fp16
    roundings of the same quantity differ by ~sqrt(2) x one)."""
    z, variant, sd, x, y = load_case(name)
    assert len(FP16_REF_CASES) == 6
    m = _model(variant, sd).half()
    with torch.no_grad():
        o = m(x.cuda().half(), y.cuda().half())
    assert o.dtype == torch.float16
    o = o.float().cpu()
    ref16 = z["out_fp16"].astype(np.float32)
    ref_err = rel_rmse(ref16, z["out_fp64"])
    our_err = rel_rmse(o, z["out_fp64"])
    print(f"{name}: HIP fp16 vs fp64 {our_err:.3e}, reference fp16 vs fp64 {ref_err:.3e}, ratio {our_err / ref_err:.3f}")
    assert 5e-4 < ref_err < 4e-3
    if x.numel() == 1:          # one pixel: one noise sample per side, not a statistic (see test_forward_half_like_reference_script)
        assert our_err <= ref_err + 2 * 2.0 ** -11, (our_err, ref_err)
        return
    assert our_err <= 1.25 * ref_err, (our_err, ref_err)
    assert rel_rmse(o, ref16) <= 2.5 * ref_err
    # fp32 master weights + fp16 compute is at least as close
    m2 = _model(variant, sd).set_compute_dtype(torch.float16)
    with torch.no_grad():
        o2 = m2(x.cuda(), y.cuda()).cpu()
    assert rel_rmse(o2, z["out_fp64"]) <= 1.25 * ref_err


def test_hipgraph_replay_equals_eager():
    """Config-0 shape (1x128x128): the captured hipGraph replays bit-identically to the eager launches,
    for new inputs too."""
    from codon_amd.graph import GraphedCODON
    sd = orc.he_state("x4", seed=17)
    m = _model("x4", sd)
    g = np.random.default_rng(2)
    mk = lambda: torch.from_numpy(g.uniform(0, 1, size=(1, 1, 128, 128)).astype(np.float32)).cuda()
    x0, y0 = mk(), mk()
    gm = GraphedCODON(m, x0, y0)
    for _ in range(3):
        x, y = mk(), mk()
        with torch.no_grad():
            ref = m(x, y)
        assert torch.equal(gm(x, y), ref)
    # a weight update after capture is refused, not replayed on stale packed weights
    assert not gm.stale()
    with torch.no_grad():
        m.conv3.weight.mul_(1.0)          # in-place, autograd-visible: bumps Tensor._version
    assert gm.stale()
    with pytest.raises(RuntimeError, match="changed after capture"):
        gm(x, y)


@pytest.mark.parametrize("variant", ["x4", "x16"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_empty_batch(variant, dtype):
    """An empty batch (a rank whose shard of a small global batch holds no image): like the reference's forward
    (tests/test_oracle.py::test_empty_batch_and_tiny_images pins that on the oracle) the module returns an empty (0,1,H,W)
    map; its backward leaves zero gradients in the 44 used parameters and None in the unused attention_*5."""
    from codon_amd import BaseNet_RMCR_fuseRMCR
    m = _model(variant, orc.he_state(variant, seed=3))
    if dtype != torch.float32:
        m.set_compute_dtype(dtype)
    e = torch.zeros((0, 1, 12, 10), device="cuda", dtype=dtype)
    with torch.no_grad():
        o = m(e, e)
    assert tuple(o.shape) == (0, 1, 12, 10) and o.dtype == dtype and o.is_cuda
    m.train()
    o = m(e, e)
    assert tuple(o.shape) == (0, 1, 12, 10) and o.requires_grad
    o.sum().backward()
    n_zero = 0
    for k, p_ in m.named_parameters():
        if k.startswith(("attention_c5", "attention_s5")):
            assert p_.grad is None, k
        else:
            assert p_.grad is not None and float(p_.grad.abs().max()) == 0.0, k
            n_zero += 1
    assert n_zero == 44
    # the next real batch is unaffected
    x = torch.rand((1, 1, 12, 10), device="cuda", dtype=dtype)
    with torch.no_grad():
        assert bool(torch.isfinite(m.eval()(x, x)).all())
        r = BaseNet_RMCR_fuseRMCR().cuda().eval()
        assert tuple(r(e.float(), e.float()).shape) == (0, 1, 12, 10)
