"""CPU oracle (numpy) for the x4/x8/x16 bicubic upsample of the synthetic-input generator --
TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu leg).

Parity vs the reference: UNPINNED -- the reference has no upsample anywhere (its depth inputs are
bicubic-upsampled offline: /root/reference/CODON_X4/test.py:70-77).  This file is the
definition; codon_amd/csrc/upsample.hip must match it bit for bit (integer index tables AND fp32
outputs); tests/test_upsample.py also cross-checks it against torch's bicubic (a=-0.75,
align_corners=False) to ~1e-6.
"""
import numpy as np


def _keys(d, a=-0.75):
    d = abs(d)
    if d <= 1.0:
        return (a + 2.0) * d ** 3 - (a + 3.0) * d ** 2 + 1.0
    if d < 2.0:
        return a * d ** 3 - 5.0 * a * d ** 2 + 8.0 * a * d - 4.0 * a
    return 0.0


def phase_weights(s):
    tab = np.zeros((s, 4))
    for r in range(s):
        t = ((2 * r + 1 - s) % (2 * s)) / (2.0 * s)
        tab[r] = [_keys(1 + t), _keys(t), _keys(1 - t), _keys(2 - t)]
    return tab.astype(np.float32)


def index_table(n, s):
    """(n*s, 4) int32 clamped source indices and (n*s,) phases for one axis."""
    dst = np.arange(n * s)
    q, r = dst // s, dst % s
    i0 = q - (2 * r + 1 < s)
    taps = np.clip(i0[:, None] - 1 + np.arange(4)[None, :], 0, n - 1).astype(np.int32)
    return taps, r.astype(np.int32)


def _dot4(w, p):
    f = np.float32
    return (f(w[0]) * p[0] + f(w[1]) * p[1]) + (f(w[2]) * p[2] + f(w[3]) * p[3])


def bicubic_upsample(lr, s):
    """lr: (B,1,h,w) float32 -> (B,1,h*s,w*s) float32; every multiply/add rounded to fp32."""
    lr = np.asarray(lr, dtype=np.float32)
    B, _, h, w = lr.shape
    wt = phase_weights(s)
    tx, rx = index_table(w, s)
    ty, ry = index_table(h, s)
    wx = wt[rx]                                   # (W,4)
    wy = wt[ry]                                   # (H,4)
    rows = lr[:, 0][:, ty, :]                     # (B,H,4,w)
    g = rows[:, :, :, tx]                         # (B,H,4,W,4): [.., k_row, x, k_col]
    hrow = (wx[None, None, None, :, 0] * g[..., 0] + wx[None, None, None, :, 1] * g[..., 1]) + \
           (wx[None, None, None, :, 2] * g[..., 2] + wx[None, None, None, :, 3] * g[..., 3])   # (B,H,4,W)
    out = (wy[None, :, 0, None] * hrow[:, :, 0] + wy[None, :, 1, None] * hrow[:, :, 1]) + \
          (wy[None, :, 2, None] * hrow[:, :, 2] + wy[None, :, 3, None] * hrow[:, :, 3])
    assert out.dtype == np.float32
    return out[:, None]
