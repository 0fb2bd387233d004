"""Thin tensor-level wrappers over the C ABI: validate device/dtype/layout, pass raw device
pointers and the CURRENT torch stream.  No arithmetic happens in Python."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    raise TypeError(f"codon_amd: unsupported dtype {t.dtype} (fp32 and bf16 only)")


def _dev(*ts):
    d = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("codon_amd: tensors must live on a HIP device (there is no CPU path)")
        if not t.is_contiguous():
            raise RuntimeError("codon_amd: tensors must be contiguous NCHW")
        if d is None:
            d = t.device
        elif t.device != d:
            raise RuntimeError("codon_amd: tensors on different devices")
    return d


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


# bench.py sets PROFILE = {"key": (ksize, cin, cout), "events": []} to bracket every launch of one conv
# variant with HIP events recorded on the launch stream (the live roofline measurement).
PROFILE = None


class Slice:
    """Channels [coff, coff+c) of a contiguous (B, ctotal, H, W) buffer."""
    __slots__ = ("buf", "coff", "c")

    def __init__(self, buf: torch.Tensor, coff: int = 0, c: Optional[int] = None):
        self.buf, self.coff = buf, coff
        self.c = buf.shape[1] - coff if c is None else c
        assert 0 <= coff and coff + self.c <= buf.shape[1]

    @property
    def ctotal(self):
        return self.buf.shape[1]

    def view(self):
        return self.buf[:, self.coff:self.coff + self.c]

    def ct(self):
        return L.Tensor(self.buf.data_ptr(), self.ctotal, self.coff)


def packed_weight(w: torch.Tensor, mode: int = L.PACK_FWD, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """Pack an OIHW fp32 conv weight into the K-major image the conv kernel streams."""
    lib = L.load()
    dev = _dev(w)
    if w.dtype != torch.float32:
        w = w.float()
    cout, cin, k, _ = w.shape
    dtype = dtype or torch.float32
    out = torch.empty(w.numel(), dtype=dtype, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.codon_conv_pack_weight(_ptr(w), _ptr(out), cout, cin, k, mode,
                                           L.F32 if dtype == torch.float32 else L.BF16, _stream(dev)),
                "conv_pack_weight")
    return out


def conv2d(x: Slice, w_packed: torch.Tensor, y: Slice, ksize: int, relu: bool = False,
           residual: Optional[Slice] = None, accumulate: bool = False):
    lib = L.load()
    dev = _dev(x.buf, w_packed, y.buf, residual.buf if residual else None)
    B, _, H, W = x.buf.shape
    assert y.buf.shape[0] == B and y.buf.shape[2:] == x.buf.shape[2:]
    flags = (L.CONV_RELU if relu else 0) | (L.CONV_ADD_RESIDUAL if residual is not None else 0) | \
            (L.CONV_ACCUM_OUT if accumulate else 0)
    d = L.ConvDesc(B, H, W, x.c, y.c, ksize, x.ctotal, x.coff, y.ctotal, y.coff,
                   residual.ctotal if residual else 0, residual.coff if residual else 0, flags, _dt(x.buf))
    if residual is not None:
        assert residual.c == y.c and residual.buf.shape[2:] == x.buf.shape[2:]
    prof = PROFILE if (PROFILE is not None and PROFILE["key"] == (ksize, x.c, y.c)) else None
    with torch.cuda.device(dev):
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        L.check(lib.codon_conv2d_fwd(C.byref(d), _ptr(x.buf), _ptr(w_packed), _ptr(y.buf),
                                     _ptr(residual.buf if residual else None), _stream(dev)), "conv2d_fwd")
        if prof is not None:
            e1.record(torch.cuda.current_stream(dev))
            prof["events"].append((e0, e1))


def stem(x: torch.Tensor, w: torch.Tensor, y: Slice):
    lib = L.load()
    dev = _dev(x, w, y.buf)
    B, _, H, W = x.shape
    assert y.c == 64 and x.dtype == torch.float32 and w.dtype == torch.float32
    with torch.cuda.device(dev):
        L.check(lib.codon_stem_fwd(B, H, W, _ptr(x), _ptr(w), _ptr(y.buf), y.ctotal, y.coff, _dt(y.buf),
                                   _stream(dev)), "stem_fwd")


def head(x: Slice, w: torch.Tensor, residual: torch.Tensor, y: torch.Tensor):
    lib = L.load()
    dev = _dev(x.buf, w, residual, y)
    B, _, H, W = x.buf.shape
    assert x.c == 64 and w.dtype == torch.float32 and residual.dtype == torch.float32 and y.dtype == torch.float32
    with torch.cuda.device(dev):
        L.check(lib.codon_head_fwd(B, H, W, _ptr(x.buf), x.ctotal, x.coff, _ptr(w), _ptr(residual), _ptr(y),
                                   _dt(x.buf), _stream(dev)), "head_fwd")


def cac_stats_tiles(H: int, W: int) -> int:
    return L.load().codon_cac_stats_tiles(H, W)


def cac_stats(pre_c: Slice, pre: Slice, pooled: torch.Tensor, partials: torch.Tensor):
    lib = L.load()
    dev = _dev(pre_c.buf, pre.buf, pooled, partials)
    B, _, H, W = pre.buf.shape
    a, b = pre_c.ct(), pre.ct()
    with torch.cuda.device(dev):
        L.check(lib.codon_cac_stats_fwd(B, H, W, C.byref(a), C.byref(b), _ptr(pooled), _ptr(partials),
                                        _dt(pre.buf), _stream(dev)), "cac_stats_fwd")


def cac_gate(B: int, H: int, W: int, partials, w1, b1, w2, b2, ch, pools_out=None):
    lib = L.load()
    dev = _dev(partials, w1, b1, w2, b2, ch, pools_out)
    for t in (w1, b1, w2, b2):
        assert t.dtype == torch.float32
    with torch.cuda.device(dev):
        L.check(lib.codon_cac_gate_fwd(B, H, W, _ptr(partials), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(ch),
                                       _ptr(pools_out), _stream(dev)), "cac_gate_fwd")


def cac_spatial(pooled: torch.Tensor, w: torch.Tensor, sp: torch.Tensor):
    lib = L.load()
    dev = _dev(pooled, w, sp)
    B, _, H, W = pooled.shape
    assert w.dtype == torch.float32
    with torch.cuda.device(dev):
        L.check(lib.codon_cac_spatial_fwd(B, H, W, _ptr(pooled), _ptr(w), _ptr(sp), _stream(dev)),
                "cac_spatial_fwd")


def cac_apply(pre: Slice, pre_c: Slice, ch, sp, inputs: Slice, inputs_c: Slice, out: Slice, out_c: Slice):
    lib = L.load()
    dev = _dev(pre.buf, pre_c.buf, ch, sp, inputs.buf, inputs_c.buf, out.buf, out_c.buf)
    B, _, H, W = pre.buf.shape
    ts = [s.ct() for s in (pre, pre_c, inputs, inputs_c, out, out_c)]
    with torch.cuda.device(dev):
        L.check(lib.codon_cac_apply_fwd(B, H, W, C.byref(ts[0]), C.byref(ts[1]), _ptr(ch), _ptr(sp),
                                        C.byref(ts[2]), C.byref(ts[3]), C.byref(ts[4]), C.byref(ts[5]),
                                        _dt(pre.buf), _stream(dev)), "cac_apply_fwd")
