"""OPT-IN split-precision convs (model.set_conv_precision("f16x3")): fp32 tensors, operands split into fp16
hi+lo, three f16 MFMAs per product, fp32 accumulate.  Must stay inside the SAME fp32 parity bar as the exact
kernels: RMSE <= 1e-4 on the network output vs the reference fixtures; per-conv rel-RMSE vs torch fp32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import codon_oracle as orc
from tests.util import GOLDEN_CASES, load_case, rel_rmse, rmse


def _rand(shape, seed, scale=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(size=shape) * scale).astype(np.float32))


@pytest.mark.parametrize("k,cin,cout", [(5, 128, 128), (5, 64, 64), (3, 64, 64), (3, 128, 64)])
@pytest.mark.parametrize("shape", [(2, 19, 45), (1, 8, 32), (1, 1, 1), (1, 33, 70)])
def test_conv_f16x3_vs_torch_fp32(k, cin, cout, shape):
    from codon_amd import _lib as L
    from codon_amd import ops
    from codon_amd.ops import Slice
    dev = torch.device("cuda:0")
    B, H, W = shape
    x = _rand((B, cin, H, W), 1)
    x[0, 0, 0, 0] = 3000.0          # large-magnitude activation: still inside fp16 range
    x[0, 1 % cin, 0, 0] = 1e-6      # tiny activation: lo part underflows, absolute error stays negligible
    w = _rand((cout, cin, k, k), 2, scale=(2.0 / (k * k * cout)) ** 0.5)
    ref = F.conv2d(x.double(), w.double(), None, 1, k // 2)
    exact32 = F.conv2d(x, w, None, 1, k // 2)
    wp = ops.packed_weight(w.to(dev), L.PACK_FWD_F16X3)
    y = torch.full((B, cout, H, W), float("nan"), device=dev)
    ops.conv2d(Slice(x.to(dev)), wp, Slice(y), k, f16x3=True)
    e_split, e_fp32 = rel_rmse(y.cpu(), ref), rel_rmse(exact32, ref)
    assert e_split < 2e-6, (e_split, e_fp32)       # fp32 itself sits at ~1e-7..3e-7 here
    r = _rand((B, cout, H, W), 3)
    ops.conv2d(Slice(x.to(dev)), wp, Slice(y), k, relu=True, f16x3=True)
    assert rel_rmse(y.cpu(), F.relu(ref)) < 2e-6
    ops.conv2d(Slice(x.to(dev)), wp, Slice(y), k, residual=Slice(r.to(dev)), f16x3=True)
    assert rel_rmse(y.cpu(), ref + r.double()) < 2e-6


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_forward_f16x3_matches_golden(name):
    from codon_amd import CODONNet, CODONNet16
    z, variant, sd, x, y = load_case(name)
    m = (CODONNet16 if variant == "x16" else CODONNet)()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval().set_conv_precision("f16x3")
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
    assert rmse(o.cpu(), z["out"]) <= 1e-4             # the same bar as the exact-fp32 path
    assert rel_rmse(o.cpu(), z["out_fp64"]) <= 5e-5
    with pytest.raises(NotImplementedError):
        m(x.cuda(), y.cuda())                            # training refuses the approximate forward


def test_forward_f16x3_config1_size():
    from codon_amd import CODONNet
    sd = orc.he_state("x4", seed=13)
    g = np.random.default_rng(3)
    x = torch.from_numpy(g.uniform(0, 1, size=(1, 1, 128, 128)).astype(np.float32))
    y = torch.from_numpy((g.integers(0, 256, size=(1, 1, 128, 128)) / 255.0).astype(np.float32))
    with torch.no_grad():
        ref = orc.forward(sd, x, y)
    m = CODONNet(); m.load_state_dict(sd); m = m.cuda().eval()
    with torch.no_grad():
        exact = m(x.cuda(), y.cuda())
        split = m.set_conv_precision("f16x3")(x.cuda(), y.cuda())
    e_exact, e_split = rmse(exact.cpu(), ref), rmse(split.cpu(), ref)
    print("rmse exact", e_exact, "split", e_split, "out std", float(ref.std()))
    assert e_split <= 1e-4 and rel_rmse(split.cpu(), ref) <= 5e-5
