"""The plain-C oracle (no PyTorch, double accumulation) against the fixtures recorded from the imported reference
and against the torch-CPU restatement: two independent restatements agreeing with the reference."""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import codon_oracle as orc
from tests.util import load_case, rmse

SMALL = ["kat0_x4_1x13x11_taps", "kat0_x4_1x5x5", "kat0_x4_1x1x1", "kat0_x4_1x3x70", "kat0_x8_1x16x16",
         "he1_x4_1x17x19", "he0_x16_1x33x9"]


@pytest.mark.parametrize("name", SMALL)
def test_c_oracle_matches_reference_fixture(name):
    z, variant, sd, x, y = load_case(name)
    o = c_oracle.forward(sd, x.numpy(), y.numpy())
    assert o.shape == z["out"].shape
    assert rmse(o, z["out_fp64"]) <= 1e-5          # closer to the reference's fp64 run than its fp32 run is
    assert rmse(o, z["out"]) <= 2e-5
    with torch.no_grad():
        t = orc.forward(sd, x, y)
    assert rmse(o, t) <= 2e-5                        # the two restatements agree


def test_c_oracle_kat0_known_answer():
    sd = orc.kat_state("x4")
    x, y = orc.kat_inputs(2, 32, 24)
    o = c_oracle.forward(sd, x.numpy(), y.numpy())
    assert abs(float(o.astype(np.float64).sum()) - 775.336777) < 2e-3
    assert np.allclose(o[0, 0, 0, :4], [0.053071, 0.377331, 0.871503, 0.236362], atol=2e-6)
