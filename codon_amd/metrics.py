"""Either side of the network in the reference's script, on device:
  postprocess_u8  clip -> *255 -> truncating uint8          /root/reference/CODON_X4/test.py:127-132
  masked_rmse     RMSE over label != 0, exact integer sums   /root/reference/CODON_X4/test.py:148-164
  ssim            ssim_exact(img1, img2)                     /root/reference/CODON_X4/ssim_2.py:36-52
  L1SSIMLoss      w_l1 * mean|p - t| + w_ssim * (1 - SSIM(p, t)) with a HIP backward (the reference ships no
                  loss -- SURVEY D8 -- so the combination is this repo's; the SSIM value is pinned)."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib as L
from . import ops


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def postprocess_u8(x: torch.Tensor) -> torch.Tensor:
    """np.clip(out, 0, 1); (out * 255).astype(np.uint8) with the product formed in the array's dtype, as numpy
    does: an fp16 network output (the reference script's default) is rounded to fp16 before the truncating cast
    (test.py:125-132).  bf16 has no numpy counterpart: it is upcast to fp32."""
    lib = L.load()
    if x.dtype not in (torch.float32, torch.float16):
        x = x.float()
    x = x.contiguous()
    dev = ops._dev(x)
    out = torch.empty(x.shape, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.codon_postprocess_u8_dt(x.numel(), _p(x), ops._dt(x), _p(out), ops._stream(dev)), "postprocess_u8")
    return out


def masked_sqerr_dev(label_u8: torch.Tensor, out_u8: torch.Tensor) -> torch.Tensor:
    """masked_rmse's two exact integer sums as a DEVICE tensor int64[2] = { sum of squared errors, count } -- no host
    synchronisation (codon_amd.infer reads them back one image later); rmse = sqrt(s / c)."""
    lib = L.load()
    assert label_u8.dtype == torch.uint8 and out_u8.dtype == torch.uint8 and out_u8.dim() == 2
    label_u8 = label_u8[:out_u8.shape[0], :out_u8.shape[1]].contiguous()
    out_u8 = out_u8.contiguous()
    dev = ops._dev(label_u8, out_u8)
    acc = torch.empty(2, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.codon_masked_sqerr(out_u8.numel(), _p(label_u8), _p(out_u8), _p(acc), ops._stream(dev)),
                "masked_sqerr")
    return acc


def masked_rmse(label_u8: torch.Tensor, out_u8: torch.Tensor) -> float:
    """test.py::EvaluationResults: label is cropped to the output's size (:151); pixels with label == 0 are
    excluded from both the error and the count."""
    s, c = (int(v) for v in masked_sqerr_dev(label_u8, out_u8).cpu())
    return math.sqrt(s / c)


def _ssim_forward(a, b, want_maps):
    lib = L.load()
    dev = ops._dev(a, b)
    assert a.shape == b.shape and a.dim() == 4 and a.shape[1] == 1 and a.dtype == torch.float32
    B, _, H, W = a.shape
    part = torch.empty(lib.codon_ssim_tiles(B, H, W), dtype=torch.float32, device=dev)
    dmaps = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev) if want_maps else None
    val = torch.empty(1, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.codon_ssim_fwd(B, H, W, _p(a), _p(b), _p(part), _p(dmaps), _p(val), ops._stream(dev)), "ssim_fwd")
    return val, dmaps


def ssim_dev(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """ssim() as a DEVICE tensor float64[1]: no host synchronisation."""
    if a.dim() == 2:
        a, b = a[None, None], b[None, None]
    v, _ = _ssim_forward(a.float().contiguous(), b.float().contiguous(), False)
    return v


def ssim(a: torch.Tensor, b: torch.Tensor) -> float:
    """mean SSIM of two (B,1,H,W) or (H,W) fp32 images in [0,1] (ssim_exact's definition)."""
    return float(ssim_dev(a, b).item())


class _L1SSIMFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, w_l1, w_ssim):
        lib = L.load()
        p, t = pred.float().contiguous(), target.float().contiguous()
        dev = ops._dev(p, t)
        B, _, H, W = p.shape
        sval, dmaps = _ssim_forward(p, t, True)
        nparts = min(1024, (p.numel() + 255) // 256)
        part = torch.empty(nparts, dtype=torch.float32, device=dev)
        lval = torch.empty(1, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            L.check(lib.codon_l1_fwd(p.numel(), _p(p), _p(t), _p(part), nparts, _p(lval), ops._stream(dev)), "l1_fwd")
        ctx.save_for_backward(p, t, dmaps)
        ctx.w = (w_l1, w_ssim)
        return (w_l1 * lval + w_ssim * (1.0 - sval)).to(torch.float32).reshape(())

    @staticmethod
    def backward(ctx, g):
        lib = L.load()
        p, t, dmaps = ctx.saved_tensors
        w_l1, w_ssim = ctx.w
        dev = p.device
        B, _, H, W = p.shape
        n = p.numel()
        tmp = torch.empty_like(dmaps)
        ga = torch.empty_like(p)
        # the kernel is linear in its two scales: run it for an upstream gradient of 1 and scale the 1-channel result by
        # g ON DEVICE -- float(g) here forced a host synchronisation in every training step
        with torch.cuda.device(dev):
            L.check(lib.codon_ssim_l1_bwd(B, H, W, _p(p), _p(t), _p(dmaps), _p(tmp), _p(ga),
                                          C.c_float(-w_ssim / n), C.c_float(w_l1 / n), ops._stream(dev)),
                    "ssim_l1_bwd")
        return ga.mul_(g.to(ga.dtype)), None, None, None


class L1SSIMLoss(torch.nn.Module):
    def __init__(self, w_l1: float = 1.0, w_ssim: float = 1.0):
        super().__init__()
        self.w_l1, self.w_ssim = w_l1, w_ssim

    def forward(self, pred, target):
        return _L1SSIMFn.apply(pred, target, self.w_l1, self.w_ssim)
