"""I/O harness + checkpoint loader, and the conv-only ablation class."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import codon_oracle as orc
from tests.util import rel_rmse, rmse

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RMCR = os.path.join(ROOT, "tests", "golden", "rmcr_kat0.npz")


def _rmcr_state():
    return {k: torch.from_numpy(orc.kat_tensor(k, s)) for k, s in orc.CONV_SHAPES}


def test_oracle_rmcr_matches_reference():
    z = np.load(RMCR)
    sd = _rmcr_state()
    for nm in ("a", "b"):
        B, H, W = (int(v) for v in z[f"{nm}.shape"])
        x, y = orc.kat_inputs(B, H, W)
        with torch.no_grad():
            assert rmse(orc.forward_rmcr(sd, x, y), z[f"{nm}.out"]) < 1e-6


def test_io_roundtrip_and_checkpoint_formats(tmp_path):
    from codon_amd import CODONNet, CODONNet16, io
    img = (np.arange(37 * 53) % 256).astype(np.uint8).reshape(37, 53)
    p = str(tmp_path / "a.png")
    io.write_gray(p, img)
    assert np.array_equal(io.read_gray(p), img)
    t = io.to_input(img)
    assert t.shape == (1, 1, 37, 53) and t.dtype == torch.float32
    assert torch.equal(t, torch.from_numpy(img / 255).float()[None, None])      # test.py:122
    # the reference's checkpoint format: {"epoch", "model": whole module}; and DataParallel-prefixed dicts
    src = CODONNet()
    ck = str(tmp_path / "X4.pth")
    torch.save({"epoch": 7, "model": src}, ck)
    dst = CODONNet()
    assert io.load_checkpoint(ck, dst) == 7
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))
    s16 = CODONNet16()
    ck16 = str(tmp_path / "X16.pth")
    torch.save({"module." + k: v for k, v in s16.state_dict().items()}, ck16)
    d16 = CODONNet16()
    assert io.load_checkpoint(ck16, d16) == -1
    assert torch.equal(s16.conv3.weight, d16.conv3.weight)


@pytest.mark.gpu
def test_rmcr_ablation_matches_golden():
    from codon_amd import BaseNet_RMCR_fuseRMCR
    z = np.load(RMCR)
    m = BaseNet_RMCR_fuseRMCR()
    m.load_state_dict(_rmcr_state(), strict=True)
    m = m.cuda().eval()
    for nm in ("a", "b"):
        B, H, W = (int(v) for v in z[f"{nm}.shape"])
        x, y = orc.kat_inputs(B, H, W)
        with torch.no_grad():
            o = m(x.cuda(), y.cuda())
        assert rmse(o.cpu(), z[f"{nm}.out"]) <= 1e-4 and rel_rmse(o.cpu(), z[f"{nm}.out"]) < 1e-5
        m.set_conv_precision("f16x3")                      # opt-in split-precision convs: same bar
        with torch.no_grad():
            o3 = m(x.cuda(), y.cuda())
        m.set_conv_precision("exact")
        assert rmse(o3.cpu(), z[f"{nm}.out"]) <= 1e-4 and rel_rmse(o3.cpu(), z[f"{nm}.out"]) < 1e-5


@pytest.mark.gpu
def test_infer_cli_end_to_end(tmp_path):
    """The reference's test loop on synthetic PNGs: runs, writes outputs, prints metrics; with zeroed
    output.weight the network is the identity on the depth map, so RMSE/SSIM vs the depth itself are 0 / 1."""
    from codon_amd import CODONNet, infer, io
    g = np.random.default_rng(0)
    for d in ("depth", "color", "label", "out"):
        os.makedirs(tmp_path / d)
    for name, (h, w) in (("a.png", (40, 56)), ("b.png", (33, 47))):
        dep = g.integers(1, 256, (h, w)).astype(np.uint8)
        io.write_gray(str(tmp_path / "depth" / name), dep)
        io.write_gray(str(tmp_path / "label" / name), dep)
        io.write_gray(str(tmp_path / "color" / name), g.integers(0, 256, (h, w)).astype(np.uint8))
    m = CODONNet()
    with torch.no_grad():
        m.output.weight.zero_()
    ck = str(tmp_path / "X4.pth")
    torch.save({"epoch": 1, "model": m}, ck)
    rc = infer.main(["--scale", "4", "--input-depth", str(tmp_path / "depth"), "--input-color", str(tmp_path / "color"),
                     "--label", str(tmp_path / "label"), "--out", str(tmp_path / "out"), "--weights", ck, "--dtype", "f32"])
    assert rc == 0
    for name in ("a.png", "b.png"):
        out = io.read_gray(str(tmp_path / "out" / name))
        dep = io.read_gray(str(tmp_path / "depth" / name))
        # identity network: uint8(clip(x/255)*255) reproduces x except where float32(x/255)*255 rounds below x
        assert np.abs(out.astype(int) - dep.astype(int)).max() <= 1
