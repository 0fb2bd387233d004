"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/codon_hip.h declares (no compute without a GPU), the ctypes table mirrors the header, the
nn.Module surface matches the reference's state_dict contract, and the product path has no CPU
fallback and never imports the oracle."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "codon_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(codon_[a-z0-9_]+)\s*\(", h)))


def test_header_symbols_exported():
    from codon_amd import _lib
    names = _declared()
    assert len(names) >= 12
    assert sorted(_lib.SIGNATURES) == names, "ctypes table and include/codon_hip.h disagree"
    lib = _lib.load()
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.codon_abi_version() == _lib.ABI_VERSION
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (codon_[a-z0-9_]+)", out))
    assert set(names) <= exported


def test_host_arg_helpers_without_gpu():
    from codon_amd import _lib
    lib = _lib.load()
    assert lib.codon_cac_stats_tiles(480, 640) == 150
    assert lib.codon_cac_stats_tiles(1, 1) == 1
    assert lib.codon_conv_packed_weight_bytes(128, 128, 5, _lib.F32) == 128 * 128 * 25 * 4
    assert lib.codon_conv_packed_weight_bytes(128, 128, 4, _lib.F32) == 0
    # argument validation happens before any HIP call: null pointers -> BAD_ARG + message
    d = _lib.ConvDesc(1, 8, 8, 64, 64, 3, 64, 0, 64, 0, 0, 0, 0, _lib.F32)
    import ctypes as C
    assert lib.codon_conv2d_fwd(C.byref(d), None, None, None, None, None) == -1
    assert b"null pointer" in lib.codon_last_error_string()


def test_fp32_launch_rule_without_gpu():
    """codon_conv_tiling_f32 restates the fp32 launch rule on the host (csrc/conv_mfma_f32.hip grid_mode): a
    launch of a few rounds of workgroups is priced by its rounds -- 256 CUs, two 4-wave workgroups per CU, the last round at
    its full price -- and takes the cheapest of 8 x 32 two per CU, 4 x 32 two per CU, 4 x 32 one per CU; below 384 tiles of
    8 x 32 it is the small-grid mode (4 x 32 one per CU, or 2 x 32 cout-split up to 192 tiles alone / 128 per launch of a pair)."""
    import ctypes as C
    from codon_amd import _lib as L
    lib = L.load()

    def t(B, H, W, k=5, ci=128, co=128, chained=1, pair=0):
        d = L.ConvDesc(B, H, W, ci, co, k, ci, 0, co, 0, 0, 0, 0, L.F32)
        return lib.codon_conv_tiling_f32(C.byref(d), chained, pair)

    assert t(32, 480, 640) == L.TILING_8X32                       # the headline batch: 75 rounds, 8 x 32 wins by its 1.4 %
    assert t(16, 960, 1280) == L.TILING_8X32 and t(8, 1920, 2560) == L.TILING_8X32
    assert t(1, 370, 463) == L.TILING_4X32                        # 705 tiles: 2 pair rounds of 8 x 32 (3.945 each) vs 3 of 4 x 32 (2 each)
    assert t(1, 300, 463) == L.TILING_4X32_SOLO                   # 570 tiles: 5 solo rounds (1.093 each) beat 3 x 2 and 2 x 3.945
    assert t(5, 370, 463) == L.TILING_8X32                        # 3 525 tiles: 7 x 3.945 = 27.6 < 14 x 2
    assert t(1, 480, 640) == L.TILING_4X32 and t(1, 440, 463) == L.TILING_4X32_SOLO and t(1, 480, 463) == L.TILING_8X32
    assert t(1, 247, 343) == L.TILING_4X32_SOLO                   # 341 tiles of 8 x 32: small-grid mode, 682 tiles of 4 x 32
    assert t(1, 128, 128) == L.TILING_2X32_COUT_SPLIT and t(1, 128, 128, pair=1) == L.TILING_2X32_COUT_SPLIT
    assert t(1, 128, 160) == L.TILING_2X32_COUT_SPLIT and t(1, 128, 160, pair=1) == L.TILING_4X32_SOLO     # 160 tiles: > 128 per pair launch
    assert t(1, 128, 200) == L.TILING_4X32_SOLO                   # 224 tiles of 4 x 32 > 192
    # plain convs: 5x5 by the same pricing (its own solo constant), 3x3 / 1x1 only small-grid or not
    assert t(1, 370, 463, 5, 64, 64, 0) == L.TILING_4X32 and t(32, 480, 640, 5, 64, 64, 0) == L.TILING_8X32
    assert t(1, 370, 463, 3, 64, 64, 0) == L.TILING_8X32 and t(1, 128, 128, 3, 64, 64, 0) == L.TILING_2X32_COUT_SPLIT
    assert t(1, 128, 128, 1, 128, 64, 0) == L.TILING_4X32_SOLO    # the 1x1 conv has no cout-split form
    assert lib.codon_conv_tiling_f32(None, 0, 0) == -1 and b"conv_tiling_f32" in lib.codon_last_error_string()


def test_module_surface_matches_reference_contract(golden_dir):
    from codon_amd import CODONNet, CODONNet16
    ref = {}
    for ln in open(os.path.join(golden_dir, "state_dict_keys.txt")):
        v, k, s = ln.split()
        ref.setdefault(v, []).append((k, tuple(int(d) for d in s.split("x"))))
    m4, m16 = CODONNet(), CODONNet16()
    assert [(k, tuple(v.shape)) for k, v in m4.state_dict().items()] == ref["x4"] == ref["x8"]
    assert [(k, tuple(v.shape)) for k, v in m16.state_dict().items()] == ref["x16"]
    assert sum(p.numel() for p in m4.parameters()) == 1866136
    assert sum(p.numel() for p in m16.parameters()) == 1865506
    # He init rule for the main convs (CODON_x4.py:50-53): std = sqrt(2/(k*k*cout))
    w = m4.conv3.weight
    assert abs(float(w.detach().std()) - (2.0 / (25 * 128)) ** 0.5) < 2e-4
    # survives the things the reference scripts do to it
    import copy
    import io
    import pickle
    m4.half().float().eval().train()
    copy.deepcopy(m4)
    buf = io.BytesIO()
    torch.save({"epoch": 3, "model": m4}, buf)            # the reference's checkpoint format (test.py:56-59)
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)
    m4.load_state_dict(ck["model"].state_dict(), strict=True)
    from codon_amd import strip_module_prefix
    m16.load_state_dict(strip_module_prefix({"module." + k: v for k, v in m16.state_dict().items()}), strict=True)


def test_compat_module_names():
    code = ("import sys; sys.path.insert(0, %r); import CODON_x4, CODON_x8, CODON_x16, CAC_module; "
            "assert len(CODON_x4.CODONNet().state_dict()) == 49; assert len(CODON_x8.CODONNet().state_dict()) == 49; "
            "assert len(CODON_x16.CODONNet().state_dict()) == 44; print('ok')") % os.path.join(ROOT, "codon_amd", "compat")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_no_cpu_fallback_and_no_oracle_in_product():
    from codon_amd import CODONNet
    m = CODONNet()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 8, 8), torch.zeros(1, 3, 8, 8))
    for dp, _, fs in os.walk(os.path.join(ROOT, "codon_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "F.conv2d" not in src and "nn.functional.conv2d" not in src, f


def test_integration_doc_names_every_entry_point():
    """INTEGRATION.md's table ("Entry point | Replaces (reference file:line)") covers the whole header: a new entry point
    without a stated counterpart in the reference fails here."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "codon_hip.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    fns = sorted(set(re.findall(r"\b(codon_[a-z0-9_]+)\s*\(", header)))
    assert len(fns) >= 45
    missing = [f for f in fns if f not in doc]
    assert not missing, missing
