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
    if t.dtype == torch.float16:
        return L.F16
    raise TypeError(f"codon_amd: unsupported dtype {t.dtype} (fp32, bf16, fp16)")


_DT_CODE = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}


def _dev(*ts):
    d = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("codon_amd: tensors must live on a HIP device (there is no CPU path)")
        if not t.is_contiguous():
            raise RuntimeError("codon_amd: tensors must be contiguous")
        if d is None:
            d = t.device
        elif t.device != d:
            raise RuntimeError("codon_amd: tensors on different devices")
    return d


def _stream(dev) -> int:
    # the raw handle of torch's current stream on `dev` (torch.cuda.current_stream(dev).cuda_stream builds a Stream object and
    # resolves the device three times: 4 us per launch of an eager one-image forward); ctypes takes the int for a void*
    return torch._C._cuda_getCurrentRawStream(dev.index if dev.index is not None else torch.cuda.current_device())


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def _on(dev):
    """`with _on(dev):` = torch.cuda.device(dev), or nothing at all when `dev` already is the current device (the usual case:
    the context manager costs 3-4 us per launch, which at one small image per call is the forward's host time)."""
    i = dev.index
    return _NULL if (i is None or i == torch.cuda.current_device()) else torch.cuda.device(dev)


def _ptr(t: Optional[torch.Tensor]):
    return t.data_ptr() if t is not None else None        # ctypes converts int / None for a c_void_p argument


# bench.py sets PROFILE = {"key": (ksize, cin, cout), "events": []} to bracket every launch of one conv
# variant with HIP events recorded on the launch stream (the live roofline measurement); "wgrad_key" /
# "wgrad_events" do the same for one weight-gradient shape (the wgrad launch + its fixed-order reduce).
PROFILE = None


def is_c8(dtype: torch.dtype) -> bool:
    """16-bit activations are stored channel-blocked, [B][C/8][H][W][8] (csrc/c8.h); fp32 ones NCHW."""
    return dtype in (torch.bfloat16, torch.float16)


def new_act(B: int, C: int, H: int, W: int, dtype: torch.dtype, device) -> torch.Tensor:
    """Uninitialised activation buffer of C channels in the layout the kernels use for `dtype`."""
    if is_c8(dtype):
        assert C % 8 == 0
        return torch.empty((B, C // 8, H, W, 8), dtype=dtype, device=device)
    return torch.empty((B, C, H, W), dtype=dtype, device=device)


def from_nchw(t: torch.Tensor, dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """(B,C,H,W) tensor -> activation buffer of `dtype` (a layout change for the 16-bit types; test / tool plumbing)."""
    dtype = dtype or t.dtype
    t = t.to(dtype)
    if not is_c8(dtype):
        return t.contiguous()
    B, C, H, W = t.shape
    return t.reshape(B, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous()


def to_nchw(buf: torch.Tensor) -> torch.Tensor:
    """Activation buffer -> (B,C,H,W) tensor (a copy for the channel-blocked 16-bit layout)."""
    if buf.dim() == 4:
        return buf
    B, CB, H, W, _ = buf.shape
    return buf.permute(0, 1, 4, 2, 3).reshape(B, CB * 8, H, W)


def _bhw(buf: torch.Tensor):
    return buf.shape[0], buf.shape[2], buf.shape[3]


def _channels(buf: torch.Tensor) -> int:
    return buf.shape[1] * 8 if buf.dim() == 5 else buf.shape[1]


class Slice:
    """Channels [coff, coff+c) of an activation buffer: contiguous (B, ctotal, H, W) for fp32, channel-blocked
    (B, ctotal/8, H, W, 8) for bf16 / fp16 (then coff and c are multiples of 8)."""
    __slots__ = ("buf", "coff", "c", "ctotal")

    def __init__(self, buf: torch.Tensor, coff: int = 0, c: Optional[int] = None):
        sh = buf.shape
        blocked = len(sh) == 5
        ct = sh[1] * 8 if blocked else sh[1]
        if c is None:
            c = ct - coff
        self.buf, self.coff, self.c, self.ctotal = buf, coff, c, ct
        # (one combined test: ~115 slices are built per forward, and at one small image per call that is host time)
        half = buf.dtype is torch.bfloat16 or buf.dtype is torch.float16
        if coff < 0 or coff + c > ct or (blocked and (sh[4] != 8 or not half or (coff | c) & 7)) or \
                (not blocked and (len(sh) != 4 or half)):
            assert 0 <= coff and coff + c <= ct, "channel slice outside its buffer"
            assert blocked or not half, "16-bit activations must be channel-blocked (ops.from_nchw)"
            assert False, "a 16-bit slice is whole 8-channel planes of a (B, C/8, H, W, 8) buffer; fp32 buffers are (B, C, H, W)"

    def view(self):
        """The slice as a (B,c,H,W) tensor (a copy for the channel-blocked layout)."""
        return to_nchw(self.buf)[:, self.coff:self.coff + self.c]

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
    with _on(dev):
        L.check(lib.codon_conv_pack_weight(_ptr(w), _ptr(out), cout, cin, k, mode, _DT_CODE[dtype], _stream(dev)),
                "conv_pack_weight")
    return out


class conv_pair:
    """`with ops.conv_pair(dev):` -- the (up to two, mutually independent) conv calls of the body leave as ONE launch when
    they run the same kernel on the same grid (codon_conv_pair_begin / _end); `.launches` = how many launches it took."""

    def __init__(self, dev, enabled: bool = True):
        self.dev, self.enabled, self.launches = dev, enabled, None

    def __enter__(self):
        if self.enabled:
            L.check(L.load().codon_conv_pair_begin(), "conv_pair_begin")
        return self

    def __exit__(self, et, ev, tb):
        if not self.enabled:
            return False
        with _on(self.dev):
            n = L.load().codon_conv_pair_end(_stream(self.dev))
        if n < 0 and et is None:
            L.check(n, "conv_pair_end")
        self.launches = n
        return False


def conv2d(x: Slice, w_packed: torch.Tensor, y: Slice, ksize: int, relu: bool = False,
           residual: Optional[Slice] = None, accumulate: bool = False, relu_mask: Optional[Slice] = None,
           f16x3: bool = False, mask_sum: bool = False):
    """y = conv(x) [relu] [+ residual] ; relu_mask: y = (relu_mask > 0) ? conv(x) : 0 (backward through a
    ReLU given its output); accumulate: y += result; mask_sum (with relu_mask and accumulate): the mask applies to the
    sum, y = (relu_mask > 0) ? conv(x) + y : 0."""
    if relu_mask is not None:
        assert residual is None
        residual = relu_mask
    lib = L.load()
    dev = _dev(x.buf, w_packed, y.buf, residual.buf if residual else None)
    B, H, W = _bhw(x.buf)
    assert _bhw(y.buf) == (B, H, W)
    flags = (L.CONV_RELU if relu else 0) | (L.CONV_ACCUM_OUT if accumulate else 0) | (L.CONV_F16X3 if f16x3 else 0)
    if mask_sum:
        assert relu_mask is not None and accumulate and not relu
        flags |= L.CONV_MASK_SUM
    if relu_mask is not None:
        flags |= L.CONV_MASK_RELU
    elif residual is not None:
        flags |= L.CONV_ADD_RESIDUAL
    d = L.ConvDesc(B, H, W, x.c, y.c, ksize, x.ctotal, x.coff, y.ctotal, y.coff,
                   residual.ctotal if residual else 0, residual.coff if residual else 0, flags, _dt(x.buf))
    if residual is not None:
        assert residual.c == y.c and _bhw(residual.buf) == (B, H, W)
    prof = PROFILE if (PROFILE is not None and PROFILE["key"] == (ksize, x.c, y.c)) else None
    with _on(dev):
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        L.check(lib.codon_conv2d_fwd(C.byref(d), _ptr(x.buf), _ptr(w_packed), _ptr(y.buf),
                                     _ptr(residual.buf if residual else None), _stream(dev)), "conv2d_fwd")
        if prof is not None:
            e1.record(torch.cuda.current_stream(dev))
            prof["events"].append((e0, e1))


def conv2d_sum_into(x: Slice, w_packed: torch.Tensor, y: Slice, ksize: int, total: Slice, accumulate: bool = False):
    """16-bit, conv5x5 64 -> 64: y (+)= conv(x) and, in the same epilogue, total += y (the stored value): the last input
    gradient that fans into a block's dL/d(out) also feeds the running dL/d(inputs) (codon_conv2d_sum_into_fwd)."""
    lib = L.load()
    dev = _dev(x.buf, w_packed, y.buf, total.buf)
    B, H, W = _bhw(x.buf)
    assert _bhw(y.buf) == (B, H, W) == _bhw(total.buf) and total.c == y.c and is_c8(x.buf.dtype)
    assert total.buf.dtype == y.buf.dtype == x.buf.dtype
    for other in (y, x):                  # total may be another channel slice of y's or x's buffer, never an overlapping one
        assert total.buf.data_ptr() != other.buf.data_ptr() or (
            total.coff + total.c <= other.coff or other.coff + other.c <= total.coff)
    d = L.ConvDesc(B, H, W, x.c, y.c, ksize, x.ctotal, x.coff, y.ctotal, y.coff, total.ctotal, total.coff,
                   L.CONV_ACCUM_OUT if accumulate else 0, _dt(x.buf))
    prof = PROFILE if (PROFILE is not None and PROFILE["key"] == (ksize, x.c, y.c)) else None
    with _on(dev):
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        L.check(lib.codon_conv2d_sum_into_fwd(C.byref(d), _ptr(x.buf), _ptr(w_packed), _ptr(y.buf), _ptr(total.buf),
                                              _stream(dev)), "conv2d_sum_into_fwd")
        if prof is not None:
            e1.record(torch.cuda.current_stream(dev))
            prof["events"].append((e0, e1))


def conv_chain1x1(x: Slice, w_packed: torch.Tensor, w_chain: torch.Tensor, out: Slice, mid: Optional[Slice] = None,
                  residual: Optional[Slice] = None, f16x3: bool = False, stats=None):
    """out = conv1x1(relu(conv5x5(x))) [+ residual] in one launch (the 1x1 runs from the 5x5's accumulators);
    mid, when given, also receives relu(conv5x5(x)) (training saves it).  stats = (pool (B,2,H,W), partials
    (B, cac_fused_parts(H, W, dtype), 128, 2), choff in {0, 64}): the launch also leaves the CAC statistics of its 64
    output channels (16-bit: csrc/conv_c8.hip, per-tile partials; fp32: per-row-strip partials, tiling-invariant; finished
    by cac_tail, or cac_fused_finish + cac_gate_folded)."""
    lib = L.load()
    dev = _dev(x.buf, w_packed, w_chain, out.buf, mid.buf if mid else None, residual.buf if residual else None,
               stats[0] if stats else None, stats[1] if stats else None)
    B, H, W = _bhw(x.buf)
    assert out.c == 64 and _bhw(out.buf) == (B, H, W)
    assert out.buf.dtype == x.buf.dtype and (mid is None or (mid.c == 128 and _bhw(mid.buf) == (B, H, W)))
    assert residual is None or (residual.c == 64 and _bhw(residual.buf) == (B, H, W))
    flags = L.CONV_RELU | (L.CONV_F16X3 if f16x3 else 0)
    d = L.ConvDesc(B, H, W, x.c, 128, 5, x.ctotal, x.coff, mid.ctotal if mid else 128, mid.coff if mid else 0,
                   0, 0, flags, _dt(x.buf))
    prof = PROFILE if (PROFILE is not None and PROFILE["key"] == (5, x.c, 128)) else None
    with _on(dev):
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        ot, rt = out.ct(), (residual.ct() if residual else None)
        if stats is not None:
            pool, partials, choff = stats
            assert pool.dtype == torch.float32 and tuple(pool.shape) == (B, 2, H, W)
            assert partials.dtype == torch.float32 and tuple(partials.shape) == (B, cac_fused_parts(H, W, x.buf.dtype), 128, 2)
            L.check(lib.codon_conv_chain1x1_stats_fwd(C.byref(d), _ptr(x.buf), _ptr(w_packed),
                                                      _ptr(mid.buf if mid else None), _ptr(w_chain), C.byref(ot),
                                                      C.byref(rt) if rt is not None else None, _ptr(pool), _ptr(partials),
                                                      choff, _stream(dev)), "conv_chain1x1_stats_fwd")
        else:
            L.check(lib.codon_conv_chain1x1_fwd(C.byref(d), _ptr(x.buf), _ptr(w_packed), _ptr(mid.buf if mid else None),
                                                _ptr(w_chain), C.byref(ot), C.byref(rt) if rt is not None else None,
                                                _stream(dev)), "conv_chain1x1_fwd")
        if prof is not None:
            e1.record(torch.cuda.current_stream(dev))
            prof["events"].append((e0, e1))
            prof["chained"] = True


def conv2d_gated(pre: Slice, inputs: Slice, ch: torch.Tensor, sp: torch.Tensor, w_packed: torch.Tensor, y: Slice,
                 ksize: int, relu: bool = False, emit: Optional[Slice] = None):
    """y = conv(pre * (ch * sp) + inputs) [relu]: the CAC gate-apply of the producing block formed while staging.
    emit (16-bit): the gated input itself is also written there, for the sibling conv on the same input."""
    lib = L.load()
    dev = _dev(pre.buf, inputs.buf, ch, sp, w_packed, y.buf)
    B, H, W = _bhw(pre.buf)
    assert inputs.c == pre.c and _bhw(inputs.buf) == (B, H, W)
    assert ch.dtype == torch.float32 and tuple(ch.shape) == (B, 64) and ch.is_contiguous()
    assert sp.dtype == torch.float32 and tuple(sp.shape) == (B, 1, H, W) and sp.is_contiguous()
    d = L.ConvDesc(B, H, W, pre.c, y.c, ksize, pre.ctotal, pre.coff, y.ctotal, y.coff, 0, 0,
                   L.CONV_RELU if relu else 0, _dt(pre.buf))
    it = inputs.ct()
    with _on(dev):
        if emit is not None:
            assert emit.c == pre.c and _bhw(emit.buf) == (B, H, W) and emit.buf.dtype == pre.buf.dtype
            et = emit.ct()
            L.check(lib.codon_conv2d_gated_emit_fwd(C.byref(d), _ptr(pre.buf), C.byref(it), _ptr(ch), _ptr(sp),
                                                    _ptr(w_packed), _ptr(y.buf), C.byref(et), _stream(dev)),
                    "conv2d_gated_emit_fwd")
        else:
            L.check(lib.codon_conv2d_gated_fwd(C.byref(d), _ptr(pre.buf), C.byref(it), _ptr(ch), _ptr(sp), _ptr(w_packed),
                                               _ptr(y.buf), _stream(dev)), "conv2d_gated_fwd")


class DeferredReduce:
    """Every fixed-order reduction of a backward pass as ONE launch (codon_reduce_multi).  The producers (weight-gradient
    kernels, the CAC gate / spatial backward, the 1-channel weight gradients) are called with `defer=(this, key)`: they
    leave their per-split partials in workspaces this object keeps alive, and run(outs) reduces them all -- the up to five
    uses of a shared weight in the order they were produced -- writing or ADDING into outs[key].  Bit for bit the values the
    immediate form (one small reduce launch behind each producer) leaves."""

    def __init__(self):
        self.wg = {}        # key -> dict(cout, cin, taps, nparts, ws=[...])
        self.rows = []      # (key, part tensor, element offset, n, nparts, stride, nchunk, flip9)

    def add_wgrad(self, key, ws: torch.Tensor, cout: int, cin: int, taps: int):
        n = cout * cin * taps
        assert ws.dtype == torch.float32 and ws.numel() % n == 0
        e = self.wg.setdefault(key, dict(cout=cout, cin=cin, taps=taps, nparts=ws.numel() // n, ws=[]))
        assert (e["cout"], e["cin"], e["taps"], e["nparts"]) == (cout, cin, taps, ws.numel() // n), key
        e["ws"].append(ws)

    def add_rows(self, key, part: torch.Tensor, off: int, n: int, nparts: int, stride: int, nchunk: int = 1, flip9: bool = False):
        assert part.dtype == torch.float32 and off + (nparts - 1) * stride + n <= part.numel()
        self.rows.append((key, part, off, n, nparts, stride, nchunk, flip9))

    def keys(self):
        return list(self.wg.keys()) + [r[0] for r in self.rows]

    def run(self, outs: dict, accumulate: bool):
        """outs: key -> contiguous fp32 tensor of the gradient's size on the producers' device."""
        rounds = [[]]               # one launch per round; a weight with more than 5 uses continues in the next round
        for key, e in self.wg.items():
            o = outs[key]
            assert o.dtype == torch.float32 and o.is_contiguous() and o.numel() == e["cout"] * e["cin"] * e["taps"], key
            uses = e["ws"]
            for r, u0 in enumerate(range(0, len(uses), L.REDUCE_MAX_USES)):
                it = L.ReduceItem()
                it.out = o.data_ptr()
                chunk = uses[u0:u0 + L.REDUCE_MAX_USES]
                for j, w_ in enumerate(chunk):
                    it.part[j] = w_.data_ptr()
                it.nuse, it.nparts, it.cout, it.cin, it.taps, it.nchunk = len(chunk), e["nparts"], e["cout"], e["cin"], e["taps"], 1
                it.flags = L.REDUCE_WGRAD | (L.REDUCE_ACCUMULATE if (accumulate or u0 > 0) else 0)
                while len(rounds) <= r:
                    rounds.append([])
                rounds[r].append(it)       # (items of ONE launch run concurrently: two of them must never share `out`)
        for key, part, off, n, nparts, stride, nchunk, flip9 in self.rows:
            o = outs[key]
            assert o.dtype == torch.float32 and o.is_contiguous() and o.numel() == n, key
            it = L.ReduceItem()
            it.out = o.data_ptr()
            it.part[0] = part.data_ptr() + 4 * off
            it.stride, it.nuse, it.nparts, it.cout, it.cin, it.taps, it.nchunk = stride, 1, nparts, 1, n, 1, nchunk
            it.flags = (L.REDUCE_ACCUMULATE if accumulate else 0) | (L.REDUCE_FLIP9 if flip9 else 0)
            rounds[0].append(it)
        if not rounds[0]:
            return
        dev = _dev(*[outs[k] for k in self.keys()])
        with _on(dev):
            for items in rounds:
                arr = (L.ReduceItem * len(items))(*items)
                L.check(L.load().codon_reduce_multi(arr, len(items), _stream(dev)), "reduce_multi")
        self.wg, self.rows = {}, []


def conv2d_wgrad(x: Slice, gy: Slice, dw: Optional[torch.Tensor], ksize: int, accumulate: bool = False, defer=None):
    """dw (cout,cin,k,k) fp32 (+)= dL/dw of y = conv(x, w) given gy = dL/dy.  defer = (DeferredReduce, key): the splits
    stay in the workspace and are reduced by that object's one launch; dw is not used."""
    lib = L.load()
    dev = _dev(x.buf, gy.buf, dw)
    B, H, W = _bhw(x.buf)
    assert defer is not None or (dw.dtype == torch.float32 and tuple(dw.shape) == (gy.c, x.c, ksize, ksize))
    d = L.ConvDesc(B, H, W, x.c, gy.c, ksize, x.ctotal, x.coff, gy.ctotal, gy.coff, 0, 0, 0, _dt(x.buf))
    nbytes = lib.codon_conv_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError(f"codon_amd: no wgrad kernel for k={ksize} cin={x.c} cout={gy.c}")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    prof = PROFILE if (PROFILE is not None and PROFILE.get("wgrad_key") == (ksize, x.c, gy.c)) else None
    with _on(dev):
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
        mode = L.WGRAD_DEFER if defer is not None else (1 if accumulate else 0)
        L.check(lib.codon_conv2d_wgrad(C.byref(d), _ptr(x.buf), _ptr(gy.buf), None if defer is not None else _ptr(dw),
                                       _ptr(ws), nbytes, mode, _stream(dev)), "conv2d_wgrad")
        if prof is not None:
            e1.record(torch.cuda.current_stream(dev))
            prof["wgrad_events"].append((e0, e1))
    if defer is not None:
        defer[0].add_wgrad(defer[1], ws, gy.c, x.c, ksize * ksize)


def conv1x1_bwd(x: Slice, gy: Slice, w_packed_dgrad: torch.Tensor, gx: Slice, dw: Optional[torch.Tensor],
                accumulate: bool = False, defer=None):
    """16-bit, 1x1 conv 128 -> 64 on a ReLU output x: dw (+)= dL/dw and gx = (W^T gy) * [x > 0] in one pass over x, gy.
    defer: as conv2d_wgrad."""
    lib = L.load()
    dev = _dev(x.buf, gy.buf, w_packed_dgrad, gx.buf, dw)
    B, H, W = _bhw(x.buf)
    assert defer is not None or (dw.dtype == torch.float32 and tuple(dw.shape) == (gy.c, x.c, 1, 1))
    assert gx.c == x.c and _bhw(gx.buf) == (B, H, W)
    assert gx.buf.dtype == x.buf.dtype == gy.buf.dtype and gx.buf.data_ptr() not in (x.buf.data_ptr(), gy.buf.data_ptr())
    d = L.ConvDesc(B, H, W, x.c, gy.c, 1, x.ctotal, x.coff, gy.ctotal, gy.coff, 0, 0, 0, _dt(x.buf))
    nbytes = lib.codon_conv_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError(f"codon_amd: no wgrad kernel for k=1 cin={x.c} cout={gy.c}")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    gt = gx.ct()
    with _on(dev):
        mode = L.WGRAD_DEFER if defer is not None else (1 if accumulate else 0)
        L.check(lib.codon_conv1x1_bwd(C.byref(d), _ptr(x.buf), _ptr(gy.buf), _ptr(w_packed_dgrad), C.byref(gt),
                                      None if defer is not None else _ptr(dw), _ptr(ws), nbytes, mode, _stream(dev)),
                "conv1x1_bwd")
    if defer is not None:
        defer[0].add_wgrad(defer[1], ws, gy.c, x.c, 1)


def stem(x: torch.Tensor, w: torch.Tensor, y: Slice):
    lib = L.load()
    dev = _dev(x, w, y.buf)
    B, _, H, W = x.shape
    assert y.c == 64 and x.dtype == torch.float32 and w.dtype == torch.float32
    with _on(dev):
        L.check(lib.codon_stem_fwd(B, H, W, _ptr(x), _ptr(w), _ptr(y.buf), y.ctotal, y.coff, _dt(y.buf),
                                   _stream(dev)), "stem_fwd")


def stem_pair(xa: torch.Tensor, wa: torch.Tensor, ya: Slice, xb: torch.Tensor, wb: torch.Tensor, yb: Slice):
    """stem(xa, wa, ya) and stem(xb, wb, yb) as one launch (codon_stem_pair_fwd): same bits."""
    lib = L.load()
    dev = _dev(xa, wa, ya.buf, xb, wb, yb.buf)
    B, _, H, W = xa.shape
    assert xb.shape == xa.shape and ya.c == 64 and yb.c == 64 and ya.buf.dtype == yb.buf.dtype
    assert all(t.dtype == torch.float32 for t in (xa, wa, xb, wb))
    with _on(dev):
        L.check(lib.codon_stem_pair_fwd(B, H, W, _ptr(xa), _ptr(wa), _ptr(ya.buf), ya.ctotal, ya.coff, _ptr(xb), _ptr(wb),
                                        _ptr(yb.buf), yb.ctotal, yb.coff, _dt(ya.buf), _stream(dev)), "stem_pair_fwd")


def head(x: Slice, w: torch.Tensor, residual: torch.Tensor, y: torch.Tensor):
    lib = L.load()
    dev = _dev(x.buf, w, residual, y)
    B, H, W = _bhw(x.buf)
    assert x.c == 64 and w.dtype == torch.float32 and residual.dtype == torch.float32
    if y.dtype != torch.float32:
        # 16-bit output map of a 16-bit model (codon_head_fwd_y16): the fp32 result rounded once in the store
        assert is_c8(x.buf.dtype) and y.dtype == x.buf.dtype and y.is_contiguous()
        with _on(dev):
            L.check(lib.codon_head_fwd_y16(B, H, W, _ptr(x.buf), x.ctotal, x.coff, _ptr(w), _ptr(residual), _ptr(y),
                                           _dt(x.buf), _stream(dev)), "head_fwd_y16")
        return
    with _on(dev):
        L.check(lib.codon_head_fwd(B, H, W, _ptr(x.buf), x.ctotal, x.coff, _ptr(w), _ptr(residual), _ptr(y),
                                   _dt(x.buf), _stream(dev)), "head_fwd")


def cac_stats_tiles(H: int, W: int) -> int:
    return L.load().codon_cac_stats_tiles(H, W)


def cac_fused_tiles(H: int, W: int) -> int:
    return L.load().codon_cac_fused_tiles(H, W)


def cac_fused_parts(H: int, W: int, dtype) -> int:
    """Rows per image of the fused-statistics partials for activations of `dtype` (codon_cac_fused_parts)."""
    code = {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}[dtype]
    return L.load().codon_cac_fused_parts(H, W, code)


def params_f32(tensors):
    """The given small parameter tensors as fp32: themselves when they already are, else views of ONE flat fp32 buffer
    filled by one launch (codon_cast_multi) -- a 16-bit model's stems, head and gate tensors, read live on every call."""
    tensors = list(tensors)
    out = [t.detach() if t.dtype == torch.float32 else None for t in tensors]
    todo = [i for i, o in enumerate(out) if o is None]
    if not todo:
        return out
    lib = L.load()
    while todo:
        idx, todo = todo[:L.CAST_MAX], todo[L.CAST_MAX:]
        dev = _dev(*[tensors[i] for i in idx])
        d = L.CastDesc()
        d.n = len(idx)
        for j, i in enumerate(idx):
            t = tensors[i]
            d.src[j], d.count[j], d.dtype[j] = t.data_ptr(), t.numel(), _dt(t)
        flat = torch.empty(sum(tensors[i].numel() for i in idx), dtype=torch.float32, device=dev)
        with _on(dev):
            L.check(lib.codon_cast_multi(C.byref(d), _ptr(flat), _stream(dev)), "cast_multi")
        off = 0
        for i in idx:
            n = tensors[i].numel()
            out[i] = flat[off:off + n].view(tensors[i].shape)
            off += n
    return out


def cac_tail(B: int, H: int, W: int, partials, pool_c, pool_d, pooled, folded, counters, w1, b1, w2, b2, ws, ch, sp,
             pools_out=None):
    """The whole gate of a block in one launch (codon_cac_tail_fwd): pool_c / pool_d given = the 16-bit path's two per-stream
    maps (pooled, when not None, is WRITTEN); both None = the fp32 path (pooled is READ, as cac_stats left it)."""
    lib = L.load()
    dev = _dev(partials, pool_c, pool_d, pooled, folded, counters, w1, b1, w2, b2, ws, ch, sp, pools_out)
    assert tuple(folded.shape) == (B, L.CAC_FOLDS, 128, 2) and counters.dtype == torch.int32 and counters.numel() >= B
    assert partials.shape[0] == B and tuple(partials.shape[2:]) == (128, 2)
    with _on(dev):
        L.check(lib.codon_cac_tail_fwd(B, H, W, int(partials.shape[1]), _ptr(partials), _ptr(pool_c), _ptr(pool_d), _ptr(pooled),
                                       _ptr(folded), _ptr(counters), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(ws), _ptr(ch),
                                       _ptr(pools_out), _ptr(sp), _stream(dev)), "cac_tail_fwd")


def cac_fused_finish(B: int, H: int, W: int, partials, pool_c, pool_d, folded, pooled):
    """Fold the per-tile statistics of the two conv_chain1x1(stats=...) launches and combine their per-stream maps into
    pooled (B,2,H,W) = {channel max, channel mean}."""
    lib = L.load()
    dev = _dev(partials, pool_c, pool_d, folded, pooled)
    assert tuple(folded.shape) == (B, L.CAC_FOLDS, 128, 2) and tuple(pooled.shape) == (B, 2, H, W)
    with _on(dev):
        L.check(lib.codon_cac_fused_finish(B, H, W, _ptr(partials), _ptr(pool_c), _ptr(pool_d), _ptr(folded), _ptr(pooled),
                                           _stream(dev)), "cac_fused_finish")


def cac_gate_folded(B: int, H: int, W: int, folded, w1, b1, w2, b2, ch, pools_out=None):
    lib = L.load()
    dev = _dev(folded, w1, b1, w2, b2, ch, pools_out)
    with _on(dev):
        L.check(lib.codon_cac_gate_folded_fwd(B, H, W, _ptr(folded), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(ch),
                                              _ptr(pools_out), _stream(dev)), "cac_gate_folded_fwd")


def cac_stats(pre_c: Slice, pre: Slice, pooled: torch.Tensor, partials: torch.Tensor):
    lib = L.load()
    dev = _dev(pre_c.buf, pre.buf, pooled, partials)
    B, H, W = _bhw(pre.buf)
    a, b = pre_c.ct(), pre.ct()
    with _on(dev):
        L.check(lib.codon_cac_stats_fwd(B, H, W, C.byref(a), C.byref(b), _ptr(pooled), _ptr(partials),
                                        _dt(pre.buf), _stream(dev)), "cac_stats_fwd")


def cac_stats_scaled(pre_c: Slice, pre: Slice, ch: torch.Tensor, pooled: torch.Tensor, partials: torch.Tensor):
    """ChannelPool of the channel-gated features (Fcat * ch): the spatial gate's input in the sequential-gate ablation."""
    lib = L.load()
    dev = _dev(pre_c.buf, pre.buf, ch, pooled, partials)
    B, H, W = _bhw(pre.buf)
    assert ch.dtype == torch.float32 and tuple(ch.shape) == (B, 64)
    a, b = pre_c.ct(), pre.ct()
    with _on(dev):
        L.check(lib.codon_cac_stats_scaled_fwd(B, H, W, C.byref(a), C.byref(b), _ptr(ch), _ptr(pooled), _ptr(partials),
                                               _dt(pre.buf), _stream(dev)), "cac_stats_scaled_fwd")


def ew_sq_scale(x: Slice, ch: torch.Tensor, y: Slice):
    """y = x * x * ch[b][c] (64-channel slices)."""
    lib = L.load()
    dev = _dev(x.buf, ch, y.buf)
    B, H, W = _bhw(x.buf)
    assert x.c == 64 and y.c == 64 and ch.dtype == torch.float32 and tuple(ch.shape) == (B, 64)
    xt, yt = x.ct(), y.ct()
    with _on(dev):
        L.check(lib.codon_ew_sq_scale(B, H, W, C.byref(xt), _ptr(ch), C.byref(yt), _dt(x.buf), _stream(dev)), "ew_sq_scale")


def cac_gate(B: int, H: int, W: int, partials, w1, b1, w2, b2, ch, pools_out=None):
    lib = L.load()
    dev = _dev(partials, w1, b1, w2, b2, ch, pools_out)
    for t in (w1, b1, w2, b2):
        assert t.dtype == torch.float32
    with _on(dev):
        L.check(lib.codon_cac_gate_fwd(B, H, W, _ptr(partials), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(ch),
                                       _ptr(pools_out), _stream(dev)), "cac_gate_fwd")


def cac_spatial(pooled: torch.Tensor, w: torch.Tensor, sp: torch.Tensor):
    lib = L.load()
    dev = _dev(pooled, w, sp)
    B, _, H, W = pooled.shape
    assert w.dtype == torch.float32
    with _on(dev):
        L.check(lib.codon_cac_spatial_fwd(B, H, W, _ptr(pooled), _ptr(w), _ptr(sp), _stream(dev)),
                "cac_spatial_fwd")


def cac_apply(pre: Slice, pre_c: Slice, ch, sp, inputs: Slice, inputs_c: Slice, out: Slice, out_c: Slice):
    lib = L.load()
    dev = _dev(pre.buf, pre_c.buf, ch, sp, inputs.buf, inputs_c.buf, out.buf, out_c.buf)
    B, H, W = _bhw(pre.buf)
    ts = [s.ct() for s in (pre, pre_c, inputs, inputs_c, out, out_c)]
    with _on(dev):
        L.check(lib.codon_cac_apply_fwd(B, H, W, C.byref(ts[0]), C.byref(ts[1]), _ptr(ch), _ptr(sp),
                                        C.byref(ts[2]), C.byref(ts[3]), C.byref(ts[4]), C.byref(ts[5]),
                                        _dt(pre.buf), _stream(dev)), "cac_apply_fwd")


# ---- backward -------------------------------------------------------------------------------------

def stencil_1to64(x: torch.Tensor, w: torch.Tensor, y: Slice, relu: bool = False, flip: bool = False,
                  mask: Optional[Slice] = None):
    lib = L.load()
    dev = _dev(x, w, y.buf, mask.buf if mask else None)
    B, _, H, W = x.shape
    assert y.c == 64 and w.numel() == 576 and w.dtype == torch.float32 and x.dtype == torch.float32
    yt = y.ct()
    mt = mask.ct() if mask is not None else None
    with _on(dev):
        L.check(lib.codon_stencil_1to64(B, H, W, _ptr(x), _ptr(w), C.byref(yt), (1 if relu else 0) | (2 if flip else 0),
                                        C.byref(mt) if mt is not None else None, _dt(y.buf), _stream(dev)),
                "stencil_1to64")


def conv1ch_wgrad(a: Slice, s: torch.Tensor, dw: Optional[torch.Tensor], flip: bool, accumulate: bool = False, defer=None):
    """The 64 x 9 weight gradient of the stems / the head (include/codon_hip.h); defer: as conv2d_wgrad (a rows item)."""
    lib = L.load()
    dev = _dev(a.buf, s, dw)
    B, H, W = _bhw(a.buf)
    assert a.c == 64 and s.shape[1] == 1 and (defer is not None or (dw.numel() == 576 and dw.dtype == torch.float32))
    nbytes = lib.codon_conv1ch_wgrad_workspace_bytes(B, H, W)
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    at = a.ct()
    flags = (L.W1_FLIP if flip else 0) | (L.W1_ACCUMULATE if accumulate else 0) | (L.W1_DEFER if defer is not None else 0)
    with _on(dev):
        L.check(lib.codon_conv1ch_wgrad(B, H, W, C.byref(at), _ptr(s), None if defer is not None else _ptr(dw), flags,
                                        _ptr(ws), nbytes, _dt(a.buf), _stream(dev)), "conv1ch_wgrad")
    if defer is not None:
        defer[0].add_rows(defer[1], ws, 0, 576, ws.numel() // 576, 576, nchunk=16, flip9=flip)


def ew_add_mask(dst: Slice, src: Optional[Slice] = None, mask: Optional[Slice] = None, accumulate: bool = True):
    lib = L.load()
    dev = _dev(dst.buf, src.buf if src else None, mask.buf if mask else None)
    B, H, W = _bhw(dst.buf)
    dt_, st_, mt_ = dst.ct(), (src.ct() if src else None), (mask.ct() if mask else None)
    with _on(dev):
        L.check(lib.codon_ew_add_mask(B, H, W, dst.c, C.byref(dt_), C.byref(st_) if st_ is not None else None,
                                      C.byref(mt_) if mt_ is not None else None, 1 if accumulate else 0,
                                      _dt(dst.buf), _stream(dev)), "ew_add_mask")


def ew_sum_mask(dst: Slice, srcs, mask: Optional[Slice] = None):
    """dst = mask > 0 ? sum(srcs) : 0 for 1..4 source slices (none aliasing dst); one pass for 16-bit tensors."""
    lib = L.load()
    srcs = list(srcs)
    assert 1 <= len(srcs) <= 4
    dev = _dev(dst.buf, *[s_.buf for s_ in srcs], mask.buf if mask else None)
    B, H, W = _bhw(dst.buf)
    assert all(s_.c == dst.c and _bhw(s_.buf) == (B, H, W) and s_.buf.dtype == dst.buf.dtype for s_ in srcs)
    dt_, mt_ = dst.ct(), (mask.ct() if mask else None)
    sts = [s_.ct() for s_ in srcs]
    args = [C.byref(t) for t in sts] + [None] * (4 - len(sts))
    with _on(dev):
        L.check(lib.codon_ew_sum_mask(B, H, W, dst.c, C.byref(dt_), len(sts), *args,
                                      C.byref(mt_) if mt_ is not None else None, _dt(dst.buf), _stream(dev)), "ew_sum_mask")


def _cac_param_outputs(defer, B, nsb, part_param, part_w, f32):
    """The five parameter gradients of a CAC block: fresh tensors for the immediate form, or (defer = (DeferredReduce,
    (key_w1, key_b1, key_w2, key_b2, key_ws))) rows items over the per-image / per-block partial rows."""
    if defer is None:
        return (torch.empty((8, 128), **f32), torch.empty((8,), **f32), torch.empty((64, 8), **f32), torch.empty((64,), **f32),
                torch.empty((1, 2, 5, 5), **f32))
    red, keys = defer
    for key, off, n in zip(keys[:4], (0, 1024, 1032, 1544), (1024, 8, 512, 64)):
        red.add_rows(key, part_param, off, n, B, 1608)
    red.add_rows(keys[4], part_w, 0, 50, nsb, 50, nchunk=(1 if nsb < 64 else 64))
    return (None,) * 5


def cac_backward(g_out: Slice, g_out_c: Slice, pre: Slice, pre_c: Slice, ch, sp, pooled, pools, w1, b1, w2, ws,
                 g_pre: Slice, g_pre_c: Slice, g_in: Slice, g_in_c: Slice, accumulate_in: bool, defer=None):
    """Full backward of one CAC gate block.  Returns (dw1, db1, dw2, db2, dws) fp32 tensors (None each with `defer`:
    see _cac_param_outputs)."""
    lib = L.load()
    dev = _dev(g_out.buf, g_out_c.buf, pre.buf, pre_c.buf, ch, sp, pooled, pools, w1, b1, w2, ws, g_pre.buf,
               g_pre_c.buf, g_in.buf, g_in_c.buf)
    B, H, W = _bhw(pre.buf)
    f32 = dict(dtype=torch.float32, device=dev)
    nt = lib.codon_cac_bwd_tiles(H, W)
    nsb = lib.codon_cac_bwd_spatial_blocks(B, H, W)
    g_z = torch.empty((B, 1, H, W), **f32)
    part_gch = torch.empty((B, nt, 64), **f32)
    part_arg = torch.empty((B, nt, 128), dtype=torch.int32, device=dev)
    g_pools = torch.empty((B, 2, 128), **f32)
    argpix = torch.empty((B, 128), dtype=torch.int32, device=dev)
    part_param = torch.empty((B, 1608), **f32)
    g_pooled = torch.empty((B, 2, H, W), **f32)
    part_w = torch.empty((nsb, 50), **f32)
    dw1, db1, dw2, db2, dws = _cac_param_outputs(defer, B, nsb, part_param, part_w, f32)
    t = [s.ct() for s in (g_out, g_out_c, pre, pre_c, g_pre, g_pre_c, g_in, g_in_c)]
    st = _stream(dev)
    with _on(dev):
        L.check(lib.codon_cac_bwd_reduce(B, H, W, C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), C.byref(t[3]),
                                         _ptr(ch), _ptr(sp), _ptr(pools), _ptr(g_z), _ptr(part_gch), _ptr(part_arg),
                                         _dt(pre.buf), st), "cac_bwd_reduce")
        L.check(lib.codon_cac_bwd_gate(B, H, W, _ptr(part_gch), _ptr(part_arg), _ptr(ch), _ptr(pools), _ptr(w1),
                                       _ptr(b1), _ptr(w2), _ptr(g_pools), _ptr(argpix), _ptr(part_param), _ptr(dw1),
                                       _ptr(db1), _ptr(dw2), _ptr(db2), st), "cac_bwd_gate")
        L.check(lib.codon_cac_bwd_spatial(B, H, W, _ptr(g_z), _ptr(pooled), _ptr(ws), _ptr(g_pooled), _ptr(part_w),
                                          _ptr(dws), st), "cac_bwd_spatial")
        L.check(lib.codon_cac_bwd_apply(B, H, W, C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), C.byref(t[3]),
                                        _ptr(ch), _ptr(sp), _ptr(pooled), _ptr(g_pooled), _ptr(g_pools),
                                        _ptr(argpix), C.byref(t[4]), C.byref(t[5]), C.byref(t[6]), C.byref(t[7]),
                                        1 if accumulate_in else 0, _dt(pre.buf), st), "cac_bwd_apply")
    return dw1, db1, dw2, db2, dws


def cac_backward_fused(g_out: Slice, g_out_c: Slice, pre: Slice, pre_c: Slice, ch, sp, pooled, pools, w1, b1, w2, ws,
                       g_in: Slice, g_in_c: Slice, accumulate_in: bool, defer=None):
    """16-bit tensors, training: the CAC gate backward WITHOUT the apply pass.  Pass A also records every pixel's arg-max
    channel and folds g_out into g_in (codon_cac_bwd_reduce_acc); dL/d(pre) is not materialised -- conv1x1_bwd_gated forms
    it from g_out while staging.  Returns (dw1, db1, dw2, db2, dws, gate) with gate = the operands conv1x1_bwd_gated needs.
    accumulate_in: False / 0 = g_in := g_out, True / 1 = g_in += g_out, 2 = g_in untouched (conv2d_sum_into did it)."""
    lib = L.load()
    dev = _dev(g_out.buf, g_out_c.buf, pre.buf, pre_c.buf, ch, sp, pooled, pools, w1, b1, w2, ws, g_in.buf, g_in_c.buf)
    assert is_c8(pre.buf.dtype)
    B, H, W = _bhw(pre.buf)
    f32 = dict(dtype=torch.float32, device=dev)
    i32 = dict(dtype=torch.int32, device=dev)
    nt = lib.codon_cac_bwd_tiles(H, W)
    nsb = lib.codon_cac_bwd_spatial_blocks(B, H, W)
    g_z = torch.empty((B, 1, H, W), **f32)
    part_gch = torch.empty((B, nt, 64), **f32)
    part_arg = torch.empty((B, nt, 128), **i32)
    argch = torch.empty((B, H, W), **i32)
    g_pools = torch.empty((B, 2, 128), **f32)
    argpix = torch.empty((B, 128), **i32)
    part_param = torch.empty((B, 1608), **f32)
    g_pooled = torch.empty((B, 2, H, W), **f32)
    part_w = torch.empty((nsb, 50), **f32)
    dw1, db1, dw2, db2, dws = _cac_param_outputs(defer, B, nsb, part_param, part_w, f32)
    t = [s_.ct() for s_ in (g_out, g_out_c, pre, pre_c, g_in, g_in_c)]
    st = _stream(dev)
    with _on(dev):
        L.check(lib.codon_cac_bwd_reduce_acc(B, H, W, C.byref(t[0]), C.byref(t[1]), C.byref(t[2]), C.byref(t[3]), _ptr(ch),
                                             _ptr(sp), _ptr(pools), _ptr(pooled), _ptr(g_z), _ptr(part_gch), _ptr(part_arg),
                                             _ptr(argch), C.byref(t[4]), C.byref(t[5]), int(accumulate_in),
                                             _dt(pre.buf), st), "cac_bwd_reduce_acc")
        L.check(lib.codon_cac_bwd_gate(B, H, W, _ptr(part_gch), _ptr(part_arg), _ptr(ch), _ptr(pools), _ptr(w1),
                                       _ptr(b1), _ptr(w2), _ptr(g_pools), _ptr(argpix), _ptr(part_param), _ptr(dw1),
                                       _ptr(db1), _ptr(dw2), _ptr(db2), st), "cac_bwd_gate")
        L.check(lib.codon_cac_bwd_spatial(B, H, W, _ptr(g_z), _ptr(pooled), _ptr(ws), _ptr(g_pooled), _ptr(part_w),
                                          _ptr(dws), st), "cac_bwd_spatial")
    gate = dict(ch=ch, sp=sp, g_pooled=g_pooled, g_pools=g_pools, argpix=argpix, argch=argch)
    return dw1, db1, dw2, db2, dws, gate


def conv1x1_bwd_gated(x: Slice, g_out: Slice, w_packed_dgrad: torch.Tensor, gx: Slice, dw: Optional[torch.Tensor], gate: dict,
                      fcat_base: int, accumulate: bool = False, defer=None):
    """conv1x1_bwd whose output gradient dL/d(pre) is formed from the block's dL/d(out) `g_out` (one stream's 64 channels)
    while it is staged -- see cac_backward_fused.  fcat_base: 0 = colour stream (confuse_c), 64 = depth stream (confuse)."""
    lib = L.load()
    dev = _dev(x.buf, g_out.buf, w_packed_dgrad, gx.buf, dw, *gate.values())
    B, H, W = _bhw(x.buf)
    assert defer is not None or (dw.dtype == torch.float32 and tuple(dw.shape) == (g_out.c, x.c, 1, 1))
    assert gx.c == x.c and _bhw(gx.buf) == (B, H, W)
    assert gx.buf.dtype == x.buf.dtype == g_out.buf.dtype and gx.buf.data_ptr() not in (x.buf.data_ptr(), g_out.buf.data_ptr())
    assert tuple(gate["argch"].shape) == (B, H, W) and gate["argch"].dtype == torch.int32
    d = L.ConvDesc(B, H, W, x.c, g_out.c, 1, x.ctotal, x.coff, g_out.ctotal, g_out.coff, 0, 0, 0, _dt(x.buf))
    nbytes = lib.codon_conv_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        raise RuntimeError(f"codon_amd: no wgrad kernel for k=1 cin={x.c} cout={g_out.c}")
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
    gt = gx.ct()
    with _on(dev):
        mode = L.WGRAD_DEFER if defer is not None else (1 if accumulate else 0)
        L.check(lib.codon_conv1x1_bwd_gated(C.byref(d), _ptr(x.buf), _ptr(g_out.buf), _ptr(w_packed_dgrad), C.byref(gt),
                                            None if defer is not None else _ptr(dw), _ptr(ws), nbytes, mode, _ptr(gate["ch"]),
                                            _ptr(gate["sp"]), _ptr(gate["g_pooled"]), _ptr(gate["g_pools"]),
                                            _ptr(gate["argpix"]), _ptr(gate["argch"]), fcat_base, _stream(dev)),
                "conv1x1_bwd_gated")
    if defer is not None:
        defer[0].add_wgrad(defer[1], ws, g_out.c, x.c, 1)
