"""x4/x8/x16 bicubic upsample (synthetic-input generator): oracle self-checks on CPU, bit-exact
HIP-vs-oracle parity on GPU.  Parity vs the reference is unpinned (it has no upsample)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import upsample_oracle as uo


@pytest.mark.parametrize("s", [4, 8, 16])
def test_index_tables(s):
    n = 11
    taps, ph = uo.index_table(n, s)
    dst = np.arange(n * s)
    src = (dst + 0.5) / s - 0.5                       # half-pixel centres, float statement
    i0 = np.floor(src).astype(int)
    ref = np.clip(i0[:, None] - 1 + np.arange(4)[None], 0, n - 1)
    assert np.array_equal(taps, ref)                  # integer form == floor() form, exactly
    assert np.array_equal(ph, dst % s)
    w = uo.phase_weights(s)
    assert np.allclose(w.sum(1), 1.0, atol=1e-6)      # partition of unity
    assert np.allclose(w, w[::-1, ::-1], atol=1e-7)   # phase symmetry


@pytest.mark.parametrize("s", [4, 8, 16])
def test_oracle_vs_torch_bicubic(s):
    lr = np.random.default_rng(s).uniform(0, 1, (2, 1, 9, 13)).astype(np.float32)
    o = uo.bicubic_upsample(lr, s)
    t = F.interpolate(torch.from_numpy(lr), scale_factor=s, mode="bicubic", align_corners=False).numpy()
    assert np.abs(o - t).max() < 2e-6
    const = np.full((1, 1, 3, 4), 0.625, np.float32)
    assert np.abs(uo.bicubic_upsample(const, s) - 0.625).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("s,shape", [(4, (2, 9, 13)), (8, (1, 5, 7)), (16, (1, 3, 2)), (4, (1, 1, 1)), (4, (2, 120, 160))])
def test_hip_bit_exact(s, shape):
    from codon_amd.upsample import bicubic_upsample, phase_weights
    assert np.array_equal(phase_weights(s), uo.phase_weights(s))
    B, h, w = shape
    lr = np.random.default_rng(1).uniform(0, 1, (B, 1, h, w)).astype(np.float32)
    got = bicubic_upsample(torch.from_numpy(lr).cuda(), s).cpu().numpy()
    ref = uo.bicubic_upsample(lr, s)
    assert got.shape == ref.shape
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))   # bit for bit
