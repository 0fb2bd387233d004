"""x4/x8/x16 bicubic upsample used by the synthetic-input generator (bench.py).  Host side computes
the per-phase Keys weights (fp64 -> fp32, once); the gather + arithmetic runs in upsample.hip."""
import ctypes as C
import functools

import numpy as np
import torch

from . import _lib as L
from . import ops


def _keys(d, a=-0.75):
    d = abs(d)
    if d <= 1.0:
        return (a + 2.0) * d ** 3 - (a + 3.0) * d ** 2 + 1.0
    if d < 2.0:
        return a * d ** 3 - 5.0 * a * d ** 2 + 8.0 * a * d - 4.0 * a
    return 0.0


@functools.lru_cache(maxsize=None)
def phase_weights(scale: int) -> np.ndarray:
    """(scale, 4) fp32: weights of taps i0-1..i0+2 for output phase r = dst mod scale.
    frac t = ((2r + 1 - s) mod 2s) / 2s  (integer numerator, see upsample.hip)."""
    tab = np.zeros((scale, 4), dtype=np.float64)
    for r in range(scale):
        num = (2 * r + 1 - scale) % (2 * scale)
        t = num / (2.0 * scale)
        tab[r] = [_keys(1.0 + t), _keys(t), _keys(1.0 - t), _keys(2.0 - t)]
    return tab.astype(np.float32)


def bicubic_upsample(lr: torch.Tensor, scale: int) -> torch.Tensor:
    lib = L.load()
    if lr.dim() != 4 or lr.shape[1] != 1 or lr.dtype != torch.float32:
        raise RuntimeError("bicubic_upsample expects a (B,1,h,w) fp32 tensor")
    dev = ops._dev(lr)
    B, _, h, w = lr.shape
    wt = torch.from_numpy(phase_weights(scale)).to(dev)
    out = torch.empty((B, 1, h * scale, w * scale), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.codon_bicubic_upsample(B, h, w, scale, C.c_void_p(lr.data_ptr()), C.c_void_p(wt.data_ptr()),
                                           C.c_void_p(out.data_ptr()), ops._stream(dev)), "bicubic_upsample")
    return out
