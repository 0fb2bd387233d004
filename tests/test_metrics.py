"""Metric / loss kernels: oracle pinned on CPU by fixtures recorded from the reference's own
ssim_2.py and sample PNGs; HIP parity on GPU (bit-exact for the byte/integer pieces)."""
import os

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as mo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics_crops.npz")
NAMES = ["Art", "Books", "Dolls"]


@pytest.fixture(scope="module")
def z():
    return np.load(GOLD)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_ssim_and_rmse_match_reference(z, name):
    lab, out, dep = z[f"{name}.label"], z[f"{name}.output"], z[f"{name}.depth"]
    assert abs(mo.ssim_exact(out / 255, lab / 255) - float(z[f"{name}.ssim_out_label"])) < 1e-12
    assert abs(mo.ssim_exact(dep / 255, lab / 255) - float(z[f"{name}.ssim_dep_label"])) < 1e-12
    assert abs(mo.masked_rmse(lab, out) - float(z[f"{name}.rmse_out_label"])) < 1e-12
    assert abs(mo.masked_rmse_loop(lab, dep) - float(z[f"{name}.rmse_dep_label"])) < 1e-12


def test_dataset_means_quoted_in_survey(z):
    assert abs(float(z["dataset_mean_rmse_x4"]) - 1.778) < 5e-4
    assert abs(float(z["dataset_mean_rmse_x8"]) - 3.479) < 5e-4
    assert abs(float(z["dataset_mean_rmse_x16"]) - 5.803) < 5e-4


def test_oracle_postprocess_truncates():
    x = np.array([-0.3, 0.0, 0.5, 0.999, 1.0, 1.7, 254.9 / 255], dtype=np.float32)
    assert mo.postprocess_u8(x).tolist() == [0, 0, 127, 254, 255, 255, 254]


def test_oracle_postprocess_fp16_rounds_product_to_half():
    """The reference script's default path keeps the output in float16 (test.py:52,125-132): out * 255 is rounded
    to fp16 before the truncating cast.  0.99219 (fp16) * 255 = 253.008 -> 253; 0.9917 * 255 = 252.88 -> fp16 252.875."""
    h = np.array([0.992, 0.9917, 0.5, 1.0, 0.0], dtype=np.float16)
    got = mo.postprocess_u8(h)
    assert got.dtype == np.uint8 and got.tolist() == (np.clip(h, 0, 1) * 255).astype(np.uint8).tolist()
    # there are fp16 values whose fp32 product truncates differently: the two paths are not the same function
    allh = np.arange(0, 0x3c01, dtype=np.uint16).view(np.float16)          # every fp16 in [0, 1]
    assert (mo.postprocess_u8(allh) != mo.postprocess_u8(allh.astype(np.float32))).any()


def test_ssim_torch_equals_numpy_oracle(z):
    a, b = z["Art.output"] / 255.0, z["Art.label"] / 255.0
    t = mo.ssim_torch(torch.from_numpy(a)[None, None], torch.from_numpy(b)[None, None])
    assert abs(float(t) - mo.ssim_exact(a, b)) < 1e-12


# ---- GPU ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_metrics_match_oracle(z, name):
    from codon_amd import metrics
    lab, out, dep = (torch.from_numpy(z[f"{name}.{k}"]).cuda() for k in ("label", "output", "depth"))
    assert metrics.masked_rmse(lab, out) == float(z[f"{name}.rmse_out_label"])          # exact integer sums
    assert metrics.masked_rmse(lab, dep) == float(z[f"{name}.rmse_dep_label"])
    s = metrics.ssim(out.float() / 255, lab.float() / 255)
    assert abs(s - float(z[f"{name}.ssim_out_label"])) < 2e-6                           # fp32 filter vs float64
    s = metrics.ssim(dep.float() / 255, lab.float() / 255)
    assert abs(s - float(z[f"{name}.ssim_dep_label"])) < 2e-6


@pytest.mark.gpu
def test_hip_postprocess_bit_exact():
    from codon_amd import metrics
    g = np.random.default_rng(0)
    x = np.concatenate([g.uniform(-0.2, 1.2, 100000), np.arange(0, 256) / 255.0, [0.0, 1.0, -0.0, 0.5]]).astype(np.float32)
    got = metrics.postprocess_u8(torch.from_numpy(x).cuda()).cpu().numpy()
    assert np.array_equal(got, mo.postprocess_u8(x))


@pytest.mark.gpu
def test_hip_postprocess_fp16_bit_exact_all_halves():
    """Every finite fp16 bit pattern (and the infinities): the fp16 entry equals numpy's float16 arithmetic."""
    from codon_amd import metrics
    bits = np.concatenate([np.arange(0, 0x7c01, dtype=np.uint16), np.arange(0x8000, 0xfc01, dtype=np.uint16)])
    h = bits.view(np.float16)
    got = metrics.postprocess_u8(torch.from_numpy(h.copy()).cuda()).cpu().numpy()
    assert np.array_equal(got, mo.postprocess_u8(h))
    # bf16 outputs are upcast (numpy has no bf16)
    x = torch.rand(1000, device="cuda").bfloat16()
    assert np.array_equal(metrics.postprocess_u8(x).cpu().numpy(), mo.postprocess_u8(x.float().cpu().numpy()))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(2, 40, 52), (1, 7, 7), (1, 33, 64), (3, 9, 70)])
def test_l1_ssim_loss_forward_backward(shape):
    from codon_amd.metrics import L1SSIMLoss
    B, H, W = shape
    g = np.random.default_rng(1)
    p = g.uniform(0, 1, (B, 1, H, W))
    t = np.clip(p + g.normal(0, 0.1, p.shape), 0, 1)
    pt = torch.from_numpy(p).requires_grad_(True)
    ref = (pt - torch.from_numpy(t)).abs().mean() + 0.7 * (1 - mo.ssim_torch(pt, torch.from_numpy(t)))
    ref.backward()
    pd = torch.from_numpy(p).float().cuda().requires_grad_(True)
    loss = L1SSIMLoss(1.0, 0.7)(pd, torch.from_numpy(t).float().cuda())
    loss.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-6
    gr, gh = pt.grad.float(), pd.grad.cpu()
    assert float((gh - gr).norm() / gr.norm()) < 1e-4


@pytest.mark.gpu
def test_loss_drives_the_network_backward():
    """L1+SSIM on the network output: gradients flow into CODONNet's HIP backward."""
    from codon_amd import CODONNet
    from codon_amd.metrics import L1SSIMLoss
    from oracle import codon_oracle as orc
    m = CODONNet(); m.load_state_dict(orc.he_state("x4", 2)); m = m.cuda()
    x, y = orc.kat_inputs(2, 24, 32)
    tgt = (x * 0.9 + 0.05).cuda()
    loss = L1SSIMLoss()(m(x.cuda(), y.cuda()), tgt)
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for n, p in m.named_parameters() if p.grad is not None)
    assert float(m.conv3.weight.grad.abs().sum()) > 0
