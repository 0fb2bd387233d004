"""the drop-in classes refuse what the kernels do not implement (CPU tests, one per refused argument).

The reference's gate classes take arguments the released CODONNet never varies -- CAC_channel / ChannelGate(pool_types=,
reduction_ratio=) (/root/reference/CODON_X4/CAC_module.py:27,41-56, attention/ResCBAM.py:27,40-55), BasicConv(stride=,
padding=, dilation=, groups=, relu=True by DEFAULT, bn=, bias=) (CAC_module.py:7-12).  A drop-in that accepted them and
computed avg+max / a bare conv anyway would diverge silently."""
import os
import sys

import pytest
import torch

from codon_amd import BasicConv, CAC_channel, CAC_spatial, ChannelGate, CODONNet, CODONNet16
from codon_amd.model import BaseNet_RMCR_fuseRMCR_cross


def test_default_construction_is_unchanged():
    m = CODONNet()
    assert len(m.state_dict()) == 49 and len(CODONNet16().state_dict()) == 44
    assert m.attention_c0.pool_types == ["avg", "max"] and m.attention_c5.pool_types == ["avg", "max"]
    assert m.attention_s0.spatial.bn is None and m.attention_s0.spatial.relu is None
    assert tuple(m.attention_s0.spatial.conv.weight.shape) == (1, 2, 5, 5)
    m.check_supported()
    BaseNet_RMCR_fuseRMCR_cross().check_supported()
    # the reference's own constructions (CODON_x4.py:54-65, CAC_module.py:88)
    CAC_channel(128)
    CAC_channel(128, reduction_ratio=16, pool_types=["avg", "max"])
    CAC_channel(128, pool_types=["max", "avg"])        # a two-term fp32 sum commutes: same bits
    ChannelGate(64)
    CAC_spatial()
    BasicConv(2, 1, 5, stride=1, padding=2, relu=False)


@pytest.mark.parametrize("pools", [["avg"], ["max"], ["avg", "max", "lp"], ["lse"], ["lp", "lse"], ["avg", "avg"], []])
@pytest.mark.parametrize("cls,ch", [(CAC_channel, 128), (ChannelGate, 64)])
def test_pool_types_other_than_avg_max_are_refused(cls, ch, pools):
    with pytest.raises(NotImplementedError, match="pool_types"):
        cls(ch, pool_types=pools)


@pytest.mark.parametrize("cls,args", [(CAC_channel, (64,)), (CAC_channel, (128, 8)), (CAC_channel, (256, 16)),
                                      (ChannelGate, (128,)), (ChannelGate, (64, 4))])
def test_other_gate_shapes_are_refused(cls, args):
    with pytest.raises(NotImplementedError, match="only"):
        cls(*args)


@pytest.mark.parametrize("kw", [dict(), dict(padding=2), dict(padding=2, relu=True), dict(padding=2, relu=False, bn=True),
                                dict(padding=2, relu=False, bias=True), dict(padding=2, relu=False, stride=2),
                                dict(padding=0, relu=False), dict(padding=2, relu=False, dilation=2),
                                dict(padding=2, relu=False, groups=2)])
def test_basicconv_refuses_everything_but_the_spatial_gate_conv(kw):
    """BasicConv(2, 1, 5) with the REFERENCE's defaults has padding 0 and a ReLU: it must not quietly become the bare
    padded conv."""
    with pytest.raises(NotImplementedError, match="BasicConv"):
        BasicConv(2, 1, 5, **kw)


@pytest.mark.parametrize("args", [(2, 1, 3), (2, 1, 7), (3, 1, 5), (2, 2, 5)])
def test_basicconv_refuses_other_shapes(args):
    with pytest.raises(NotImplementedError, match="BasicConv"):
        BasicConv(*args, padding=(args[2] - 1) // 2, relu=False)


def test_parameter_holders_refuse_to_be_called():
    x = torch.zeros(1, 128, 4, 4)
    for m in (CAC_channel(128), CAC_spatial(), ChannelGate(64), CAC_spatial().spatial, CAC_spatial().compress):
        with pytest.raises(NotImplementedError):
            m(x)


def test_edited_attributes_are_caught_by_the_forward_audit():
    m = CODONNet()
    m.attention_c3.pool_types = ["avg", "max", "lse"]
    with pytest.raises(NotImplementedError, match="attention_c3"):
        m.check_supported()
    m = CODONNet()
    m.attention_s1.spatial.relu = torch.nn.ReLU()
    with pytest.raises(NotImplementedError, match="attention_s1"):
        m.check_supported()
    m = CODONNet()
    m.attention_s2.spatial.bn = torch.nn.BatchNorm2d(1)
    with pytest.raises(NotImplementedError, match="attention_s2"):
        m.check_supported()
    m = CODONNet()
    m.attention_s4.spatial.conv = torch.nn.Conv2d(2, 1, 5, padding=2, bias=True)
    with pytest.raises(NotImplementedError, match="attention_s4"):
        m.check_supported()
    m = CODONNet()
    m.attention_s0.spatial.conv = torch.nn.Conv2d(2, 1, 5, padding=0, bias=False)
    with pytest.raises(NotImplementedError, match="padding"):
        m.check_supported()
    m = CODONNet()
    m.attention_c0.mlp[1] = torch.nn.Linear(128, 16)
    with pytest.raises(NotImplementedError, match="attention_c0"):
        m.check_supported()
    # attention_c5 is state only in CODONNet (never executed, CODON_x4.py:64): not audited there, audited where it runs
    m = CODONNet()
    m.attention_c5.pool_types = ["lp"]
    m.check_supported()
    c = BaseNet_RMCR_fuseRMCR_cross()
    c.attention_c5.pool_types = ["lp"]
    with pytest.raises(NotImplementedError, match="attention_c5"):
        c.check_supported()


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the read-only reference tree (build container only)")
def test_pickled_reference_module_with_other_pools_is_refused_at_the_audit(tmp_path):
    """A whole-module pickle (test.py:56-59) bypasses the constructors: a reference CODONNet whose gate was configured with
    pool_types=['avg','max','lp'] unpickles into the compat classes and must be refused by the audit every forward runs;
    the default reference module passes it."""
    import subprocess
    good, bad = str(tmp_path / "good.pth"), str(tmp_path / "bad.pth")
    code = ("import sys; sys.dont_write_bytecode = True; import torch; import CODON_x4; m = CODON_x4.CODONNet(); "
            f"torch.save({{'epoch': 1, 'model': m}}, {good!r}); m.attention_c2.pool_types = ['avg', 'max', 'lp']; "
            f"torch.save({{'epoch': 1, 'model': m}}, {bad!r})")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=os.path.join(REF, "CODON_X4"), timeout=300,
                   env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    compat = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "codon_amd", "compat")
    sys.path.insert(0, compat)
    try:
        for name in ("CODON_x4", "CAC_module", "attention", "attention.ResCBAM"):
            sys.modules.pop(name, None)
        g = torch.load(good, map_location="cpu", weights_only=False)["model"]
        assert type(g).__module__ == "codon_amd.model"
        g.check_supported()
        b = torch.load(bad, map_location="cpu", weights_only=False)["model"]
        with pytest.raises(NotImplementedError, match="attention_c2"):
            b.check_supported()
    finally:
        sys.path.remove(compat)
        for name in ("CODON_x4", "CAC_module", "attention", "attention.ResCBAM"):
            sys.modules.pop(name, None)
