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


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the read-only reference tree (build container only)")
@pytest.mark.parametrize("sub,mod,n_keys", [("CODON_X4", "CODON_x4", 49), ("CODON_X8", "CODON_x8", 49),
                                            ("CODON_X16", "CODON_x16", 44)])
def test_genuine_reference_checkpoint_unpickles(tmp_path, sub, mod, n_keys):
    """The format test.py:56-59 loads is {"epoch", "model": <the REFERENCE's CODONNet instance>}: its pickle names
    CODON_x4.CODONNet, CAC_module.* and (x4/x8) attention.ResCBAM.ChannelGate/Flatten.  Save one from the imported
    reference in a child process, load it here through codon_amd/compat."""
    import subprocess
    ck = str(tmp_path / "ref.pth")
    code = (f"import sys; sys.dont_write_bytecode = True; import torch; torch.manual_seed(5); import {mod}; "
            f"m = {mod}.CODONNet(); torch.save({{'epoch': 93, 'model': m}}, {ck!r}); "
            f"torch.save(m.state_dict(), {ck!r} + '.sd')")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=os.path.join(REF, sub), timeout=300,
                   env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    from codon_amd import CODONNet, CODONNet16, io
    dst = (CODONNet16 if n_keys == 44 else CODONNet)()
    assert io.load_checkpoint(ck, dst) == 93
    ref_sd = torch.load(ck + ".sd", map_location="cpu")
    assert list(ref_sd.keys()) == list(dst.state_dict().keys()) and len(ref_sd) == n_keys
    assert all(torch.equal(ref_sd[k], v) for k, v in dst.state_dict().items())


def test_pack_cache_invalidation_hooks():
    """writes through .data do not bump Tensor._version; invalidate_packed() / load_state_dict / _apply
    must drop the packed images (CPU-checkable: the cache dict itself)."""
    from codon_amd import CODONNet
    m = CODONNet()
    v0 = m.conv1.weight._version
    m.conv1.weight.data.normal_()
    assert m.conv1.weight._version == v0              # the hazard
    m._pack_cache["probe"] = ("tag", torch.zeros(1))
    m.invalidate_packed()
    assert not m._pack_cache
    m._pack_cache["probe"] = ("tag", torch.zeros(1))
    m.load_state_dict(m.state_dict())
    assert not m._pack_cache
    m._pack_cache["probe"] = ("tag", torch.zeros(1))
    m.double()
    assert not m._pack_cache
    m.float()


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


def test_oracle_cross_variant_reduces_to_known_pieces():
    """forward_cross restates a file that cannot be imported (parity unpinned).  What CAN be checked on the CPU: with
    the gates forced to constants it reduces to arithmetic on the pinned forward's building blocks -- sigmoid(0) = 0.5
    for every gate (zero gate weights): out = pre * 0.25 + inputs per block, fuse' = fuse^2 * 0.5 * 0.5 + fuse."""
    import torch.nn.functional as F
    sd = orc.he_state("x4", seed=5)
    for k in list(sd):
        if k.startswith("attention"):
            sd[k] = torch.zeros_like(sd[k])
    x, y = orc.kat_inputs(1, 12, 10)
    with torch.no_grad():
        got = orc.forward_cross(sd, x, y)
        r, cv, w = F.relu, orc._conv, (lambda k: sd[k + ".weight"])
        inputs = r(cv(r(cv(x, w("input"))), w("conv_input")))
        inputs_c = r(cv(r(cv(y, w("input_c"))), w("conv_input_c")))
        out, out_c = inputs, inputs_c
        for _ in range(5):
            st = torch.cat((r(cv(out, w("conv1"))), r(cv(out, w("conv2")))), 1)
            st_c = torch.cat((r(cv(out_c, w("conv4"))), r(cv(out_c, w("conv5")))), 1)
            out = cv(r(cv(st, w("conv3"))), w("confuse")) * 0.5 * 0.5 + inputs
            out_c = cv(r(cv(st_c, w("conv6"))), w("confuse_c")) * 0.5 * 0.5 + inputs_c
        fuse = r(cv(torch.cat((out, out_c), 1), w("conv7")))
        fg = fuse * (fuse * 0.5) * 0.5 + fuse
        f = fg
        for _ in range(3):
            st = torch.cat((r(cv(f, w("conv8"))), r(cv(f, w("conv9")))), 1)
            f = cv(r(cv(st, w("conv10"))), w("confuse_fuse")) + fg
        ref = cv(r(cv(f, w("conv11"))), w("output")) + x
    assert rmse(got, ref) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape,seed", [((2, 24, 40), 0), ((1, 37, 53), 1)])
def test_cross_ablation_matches_oracle_restatement(shape, seed):
    """Sequential-gate ablation on HIP vs its CPU restatement (PARITY UNPINNED: see the class docstring)."""
    from codon_amd import BaseNet_RMCR_fuseRMCR_cross
    B, H, W = shape
    sd = orc.he_state("x4", seed=30 + seed)
    g = np.random.default_rng(seed)
    x = torch.from_numpy(g.random((B, 1, H, W), dtype=np.float32))
    y = torch.from_numpy(g.random((B, 1, H, W), dtype=np.float32))
    with torch.no_grad():
        ref = orc.forward_cross(sd, x, y)
    m = BaseNet_RMCR_fuseRMCR_cross()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    with torch.no_grad():
        o = m(x.cuda(), y.cuda())
    assert rmse(o.cpu(), ref) <= 1e-4 and rel_rmse(o.cpu(), ref) < 2e-5
    mb = BaseNet_RMCR_fuseRMCR_cross()
    mb.load_state_dict(sd, strict=True)
    mb = mb.cuda().eval().set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        ob = mb(x.cuda(), y.cuda())
    assert rel_rmse(ob.cpu(), ref) <= 4e-2


@pytest.mark.gpu
def test_infer_cli_end_to_end(tmp_path, capsys):
    """The reference's test loop on synthetic PNGs: runs, writes outputs, prints metrics; with zeroed
    output.weight the network is the identity on the depth map, so RMSE/SSIM vs the depth itself are 0 / 1.
    Round 6: the loop is a three-stage pipeline (reader thread + side-stream uploads, forward, writer thread); its PNGs and
    printed metrics must be byte-identical to the reference-style serial loop (`--serial`), in fp32 and in the script's fp16."""
    from codon_amd import CODONNet, infer, io
    g = np.random.default_rng(0)
    for d in ("depth", "color", "label", "out"):
        os.makedirs(tmp_path / d)
    sizes = (("a.png", (40, 56)), ("b.png", (33, 47)), ("c.png", (64, 40)), ("d.png", (33, 47)), ("e.png", (17, 90)))
    for name, (h, w) in sizes[2:]:          # more images than the queues hold, with repeated and changing sizes
        io.write_gray(str(tmp_path / "depth" / name), g.integers(1, 256, (h, w)).astype(np.uint8))
        io.write_gray(str(tmp_path / "label" / name), g.integers(0, 256, (h + 2, w + 1)).astype(np.uint8))
        io.write_gray(str(tmp_path / "color" / name), g.integers(0, 256, (h, w + 3)).astype(np.uint8))
    for name, (h, w) in sizes[:2]:
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
    # pipeline == serial loop, byte for byte, with a network that is NOT the identity
    torch.manual_seed(5)
    ck2 = str(tmp_path / "X4b.pth")
    torch.save({"epoch": 2, "model": CODONNet()}, ck2)
    for dt in ("f32", "f16"):
        outs = {}
        for mode in ("serial", "pipe"):
            od = tmp_path / f"out_{dt}_{mode}"
            capsys.readouterr()
            rc = infer.main(["--scale", "4", "--input-depth", str(tmp_path / "depth"), "--input-color", str(tmp_path / "color"),
                             "--label", str(tmp_path / "label"), "--out", str(od), "--weights", ck2, "--dtype", dt] +
                            (["--serial"] if mode == "serial" else []))
            assert rc == 0
            outs[mode] = (capsys.readouterr().out, {n: open(od / n, "rb").read() for n, _ in sizes})
        assert outs["serial"][0] == outs["pipe"][0] and len(outs["pipe"][0].splitlines()) == 1 + len(sizes) + 2
        assert outs["serial"][1] == outs["pipe"][1]
        assert len({bytes(v) for v in outs["pipe"][1].values()}) == len(sizes)          # five different images came out


@pytest.mark.parametrize("tdt", [torch.float32, torch.float16, torch.bfloat16])
def test_infer_host_conversion_equals_the_reference_steps(tmp_path, tdt):
    """codon_amd.infer._load_host converts with numpy only (no torch CPU op inside the reader threads); the values must be
    io.to_input()'s -- float64 divide, float32 (/root/reference/CODON_X4/test.py:116-123) -- followed by the dtype's
    round-to-nearest-even, bit for bit, with both images cropped to their common size."""
    from codon_amd import infer, io
    g = np.random.default_rng(3)
    for d in ("depth", "color", "label"):
        os.makedirs(tmp_path / d)
    dep, col, lab = (g.integers(0, 256, s_).astype(np.uint8) for s_ in ((37, 53), (40, 50), (41, 55)))
    io.write_gray(str(tmp_path / "depth" / "a.png"), dep)
    io.write_gray(str(tmp_path / "color" / "a.png"), col)
    io.write_gray(str(tmp_path / "label" / "a.png"), lab)
    x, y, l_, h, w = infer._load_host(str(tmp_path / "depth"), str(tmp_path / "color"), str(tmp_path / "label"), "a.png", tdt)
    assert (h, w) == (37, 50) and x.dtype == tdt and tuple(x.shape) == (1, 1, 37, 50) == tuple(y.shape)
    assert torch.equal(x, io.to_input(dep)[:, :, :h, :w].to(tdt)) and torch.equal(y, io.to_input(col)[:, :, :h, :w].to(tdt))
    assert torch.equal(l_, torch.from_numpy(lab))
    hp = infer._pinned_copy(x) if torch.cuda.is_available() else x
    assert torch.equal(hp, x)
