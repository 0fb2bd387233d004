"""CODONNet on MI355X: the reference's nn.Module surface over hand-written HIP kernels.

Mirrors (names, constructor, forward signature, state_dict keys/shapes/order, init rule):
  /root/reference/CODON_X4/CODON_x4.py:18-132      CODONNet  (x4; CODON_X8/CODON_x8.py identical)
  /root/reference/CODON_X16/CODON_x16.py:92-202    CODONNet  (x16: no attention_c5 / attention_s5)
  /root/reference/CODON_X4/CAC_module.py:6-94      BasicConv, Flatten, CAC_channel, ChannelPool, CAC_spatial
  /root/reference/CODON_X4/attention/ResCBAM.py:26-37  ChannelGate (state only; never executed)

The sub-modules below hold PARAMETERS ONLY.  All arithmetic of forward() runs in
libcodon_hip.so (see include/codon_hip.h); there is no eager / CPU fallback -- CPU tensors raise.
"""
from __future__ import annotations

import math
import warnings
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .ops import Slice


import os as _os

# inference, 16-bit tensors: form the gate-apply in the consuming convs' staging (True) or as a pass (False); A/B switch
GATED_16BIT = _os.environ.get("CODON_GATED16", "1") != "0"
# gated inference: the conv5x5 of a sibling pair applies the gate and EMITS the gated tensor, the conv3x3 reads it as a plain
# conv (its staging is bound by the gate arithmetic otherwise: bf16 1.00 vs 0.71 ms, fp32 5.41 vs 5.05 ms at 32x480x640).
# 0 = both gated (A/B)
GATED_EMIT = _os.environ.get("CODON_GATED_EMIT", "1") != "0"
# inference on a grid too small to fill the chip (every conv launch < 256 workgroups: one 128 x 128 image is 128): the depth
# and the colour stream of a block are independent up to the CAC gate -- run them on two HIP streams.  0 = one stream (A/B)
TWO_STREAMS = _os.environ.get("CODON_TWO_STREAMS", "1") != "0"
# ... largest grid (8 x 32-pixel tiles per conv launch of one stream) that still takes the two-stream schedule: fp32 kernels /
# 16-bit kernels.  One 370 x 463 image is 705 tiles on 512 resident workgroup slots -- 1.4 rounds, i.e. two, the second a
# third full; the depth and the colour launch of a block together are 2.75 rounds (profiles/r05_b1_*)
TWO_STREAMS_MAX32 = int(_os.environ.get("CODON_TWO_STREAMS_MAX32", "256"))
TWO_STREAMS_MAX16 = int(_os.environ.get("CODON_TWO_STREAMS_MAX16", "256"))
# ... and up to this many tiles the two streams of a block leave as PAIR launches (ops.conv_pair: one grid of twice the tiles
# instead of two launches on two HIP streams -- no fork / join events, the second stream fills the first one's last round)
PAIR_MAX16 = int(_os.environ.get("CODON_PAIR_MAX16", "4096"))
# fp32: the pair form exists for the small-grid kernels (fewer than 384 tiles of 8 x 32 pixels, conv_mfma_f32.hip), where it
# replaces the two-stream schedule; 0 = two streams (A/B)
PAIR_MAX32 = int(_os.environ.get("CODON_PAIR_MAX32", "383"))
# the whole gate of a block -- pool finish, MLP, spatial conv -- in one launch (codon_cac_tail_fwd); 0 = three / four launches (A/B)
CAC_TAIL = _os.environ.get("CODON_CAC_TAIL", "1") != "0"
# fp32, images of at most 32 768 pixels: the CAC statistics out of the chained conv's epilogue; 0 = the statistics pass (A/B)
FUSED_STATS_F32 = _os.environ.get("CODON_FUSED_STATS_F32", "1") != "0"
CAC_TAIL_MAX_PIXELS = 1 << 21
_HALF_STREAMS: Dict[tuple, tuple] = {}
_TAIL_COUNTERS: Dict[tuple, torch.Tensor] = {}


def _tail_counters(dev, B: int) -> torch.Tensor:
    """The arrival counters of codon_cac_tail_fwd: zero on entry, left at zero by every launch -- so ONE zeroed buffer per
    (device, calling stream, host thread) serves every forward instead of a torch.zeros (an ATen fill launch, 5 us of a 2 ms
    one-image forward) per call.  Private to the (stream, thread) like the side streams above: launches of independent
    callers never count in each other's words; a larger batch replaces the buffer."""
    import threading
    i = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (i, torch.cuda.current_stream(dev).cuda_stream, threading.get_ident())
    t = _TAIL_COUNTERS.get(key)
    if t is None or t.numel() < B:
        if torch.cuda.is_current_stream_capturing():
            # never cache an allocation made inside a hipGraph capture (it belongs to the graph's pool)
            return torch.zeros((B,), dtype=torch.int32, device=dev)
        _prune_dead_threads(_TAIL_COUNTERS, 2)
        t = _TAIL_COUNTERS[key] = torch.zeros((max(B, 64),), dtype=torch.int32, device=dev)
    return t


def _half_chip_streams(dev, main_stream):
    """Two side streams for the two halves of a block, private to (device, calling stream, host thread): independent
    callers -- DataParallel replica threads, a hipGraph capture in one thread beside eager launches in another -- never
    share a side stream, so they get neither false cross-dependencies nor a stream that is in capture mode under them.
    (Streams created with hipExtStreamCreateWithCUMask were tried to keep the two launches on disjoint CUs: 6.0 ms
    instead of 4.4 with ANY mask, also the full one -- the external streams' event traffic; what separates the launches
    instead is the LDS request of the small-grid kernels, conv_mfma_f32.hip.)"""
    import threading
    i = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (i, main_stream.cuda_stream, threading.get_ident())
    if key not in _HALF_STREAMS:
        _prune_dead_threads(_HALF_STREAMS, 2)
        _HALF_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    return _HALF_STREAMS[key]


def _prune_dead_threads(table: dict, tid_index: int, keep: int = 16):
    """Per-(thread, stream) tables -- the side streams above, _WeightGuard.states -- would grow by one entry per short-lived
    caller thread (thread pools, nn.DataParallel's per-forward threads): before a new entry goes in, entries of threads that
    no longer exist are dropped once the table holds more than `keep`.  Their device memory returns to the caching
    allocator, which orders its reuse behind the work already enqueued on the stream it was used on."""
    if len(table) < keep:
        return
    import threading
    alive = {t.ident for t in threading.enumerate()}
    for k in [k for k in table if k[tid_index] not in alive]:
        del table[k]


# debug: re-pack on every cache hit and compare, so a write through `.data` after the first forward (the reference's own
# init idiom is m.weight.data.normal_(), CODON_x4.py:50-53) raises IN THE SAME CALL instead of one call later (_WeightGuard)
VERIFY_PACKED = _os.environ.get("CODON_VERIFY_PACKED", "0") != "0"
# the per-forward checksum launch of _WeightGuard; 0 = off (A/B of its cost only)
WEIGHT_GUARD = _os.environ.get("CODON_WEIGHT_GUARD", "1") != "0"

_STALE_MSG = ("codon_amd: a conv weight was written through `.data` (or another path that does not bump Tensor._version, "
              "e.g. `m.weight.data.normal_()`) after its packed MFMA image was built -- the forward(s) since then used the "
              "stale packed weights; call model.invalidate_packed() after such writes")


class _WeightGuard:
    """Default-on detector of stale packed weights.  The pack cache is keyed on (data_ptr, Tensor._version) of each weight,
    which a write through `.data` does not change.  Every forward launches ONE small kernel (codon_weight_checksum) over
    the raw bytes of the 17 MFMA conv weights: the first launch after the host-visible key changed records the checksum the
    packed images are built from, every later one compares and, on a mismatch, sets a flag in pinned host memory.  The
    host reads that word (no synchronisation) at the start of every forward / graph replay and in check_packed(): the
    stale forward itself has already been enqueued by then, the NEXT call raises.  Per (thread, stream) state, because
    the kernel's workspace and reference slot are ordered by the stream they are used on."""

    def __init__(self):
        self.tag = None            # host-visible key of all 17 weights the pack cache was last valid for
        self.flag = None           # pinned int32[1], written by the kernel
        self.flag_np = None
        self.states = {}           # (thread id, stream handle) -> [tag, workspace, desc, tensor whose last word is the reference]
        self.disabled = False
        self._retired = []         # flag words / workspaces of earlier epochs (see reset)

    def tripped(self) -> bool:
        return self.flag_np is not None and bool(self.flag_np[0])

    def reset(self):
        """Forget the recorded checksums and the tripped state.  A checksum launch of the stale forward may still be in
        flight and a captured hipGraph may still hold the addresses: the old flag word and workspaces are retired (kept
        alive, never reused), not cleared or freed -- a late store cannot trip the NEW flag, a replay cannot write into
        memory someone else now owns."""
        self.tag = None
        self._retired.append((self.flag, list(self.states.values())))
        del self._retired[:-8]                 # bounded: each entry is a few KB
        self.states = {}
        self.flag = None
        self.flag_np = None

    def run(self, model, dev):
        import ctypes as C
        import threading
        if self.tripped():
            raise RuntimeError(_STALE_MSG)
        ws = [getattr(model, n).weight for n in _MFMA_CONVS]
        tag = tuple((w.data_ptr(), w._version, w.dtype) for w in ws) + (dev,)
        if tag != self.tag:
            # some weight changed visibly: EVERY packed image is rebuilt, so that all of them belong to the checksum
            # recorded below (a partial rebuild could fold an earlier invisible write into the new reference)
            model._pack_cache.clear()
            self.tag = tag
            self.disabled = any((w.data_ptr() % 16) or ((w.numel() * w.element_size()) % 16) or not w.is_contiguous()
                                for w in ws)
        if self.disabled:
            return
        if self.flag is None:
            self.flag = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.flag_np = self.flag.numpy()
        stream = torch.cuda.current_stream(dev)
        key = (threading.get_ident(), stream.cuda_stream)
        st = self.states.get(key)
        lib = L.load()
        if st is None or st[1].device != dev:
            _prune_dead_threads(self.states, 0)
            n = lib.codon_weight_checksum_workspace_bytes() // 8 + 1          # + the reference slot (last word)
            st = self.states[key] = [None, torch.zeros(n, dtype=torch.int64, device=dev), None, None]
        mode = 1
        if st[0] != tag:
            d = L.WsumDesc()
            d.n = len(ws)
            for i, w in enumerate(ws):
                d.data[i] = w.data_ptr()
                d.bytes[i] = w.numel() * w.element_size()
            # under hipGraph capture a recording launch would be replayed as a recording launch and never compare: take
            # the reference another stream recorded for these very weights (GraphedCODON's warm-up runs; they are joined
            # before the capture starts) and capture a COMPARING launch
            donor = None
            if torch.cuda.is_current_stream_capturing():
                donor = next((o for o in self.states.values() if o is not st and o[0] == tag and o[1].device == dev), None)
                if donor is None:
                    del self.states[key]
                    raise RuntimeError("codon_amd: hipGraph capture of a forward whose weights no eager forward has seen "
                                       "yet -- the captured weight-checksum launch would RECORD on every replay and never "
                                       "compare; run one forward outside the capture first (GraphedCODON does: warmup >= 1)")
            st[0], st[2] = tag, d
            st[3], mode = (donor[3], 1) if donor is not None else (st[1], 0)
        ref = st[3]
        with torch.cuda.device(dev):
            L.check(lib.codon_weight_checksum(C.byref(st[2]), C.c_void_p(st[1].data_ptr()),
                                              C.c_void_p(ref.data_ptr() + 8 * (ref.numel() - 1)), mode,
                                              C.c_void_p(self.flag.data_ptr()), C.c_void_p(stream.cuda_stream)),
                    "weight_checksum")


class Conv2dParams(nn.Module):
    """Parameter holder with nn.Conv2d's attribute names (weight is OIHW, no bias)."""

    def __init__(self, in_channels, out_channels, kernel_size, he_init=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = (kernel_size, kernel_size)
        self.stride, self.padding = (1, 1), (kernel_size // 2, kernel_size // 2)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        if he_init:  # CODON_x4.py:50-53
            n = kernel_size * kernel_size * out_channels
            self.weight.data.normal_(0, math.sqrt(2.0 / n))
        else:        # nn.Conv2d default (the CAC convs are created after the He loop)
            nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))

    def forward(self, *a, **k):
        raise RuntimeError("Conv2dParams holds parameters only; CODONNet.forward runs the HIP kernels")

    def extra_repr(self):
        return f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, bias=False"


_POOLS_BUILT = ("avg", "max")


def _refuse(what: str):
    raise NotImplementedError(f"codon_amd: {what} -- the HIP kernels implement the configuration CODONNet itself uses "
                              "(CODON_x4.py:54-65) and nothing else; there is no eager fallback")


def _check_pool_types(cls: str, pool_types):
    """The reference branches on pool_types in {'avg','max','lp','lse'} (CAC_module.py:41-56, attention/ResCBAM.py:40-55);
    the kernels compute the global average AND max pool, each exactly once (their order does not matter: the two MLP
    outputs are added, and a two-term fp32 sum commutes bit for bit)."""
    pt = list(pool_types)
    if sorted(pt) != sorted(_POOLS_BUILT):
        _refuse(f"{cls}(pool_types={pt!r}): only ['avg', 'max'] is built")


class BasicConv(nn.Module):
    """CAC_module.py:6-20 / attention/ResCBAM.py:6-20, with the reference's signature AND defaults (relu=True, padding=0).
    The one configuration the kernels implement is the spatial gate's BasicConv(2, 1, 5, stride=1, padding=2, relu=False)
    (CAC_module.py:88): everything else is refused at construction instead of silently computing something different."""

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, dilation=1, groups=1, relu=True, bn=False,
                 bias=False):
        super().__init__()
        cfg = dict(in_planes=in_planes, out_planes=out_planes, kernel_size=kernel_size, stride=stride, padding=padding,
                   dilation=dilation, groups=groups, relu=relu, bn=bn, bias=bias)
        built = dict(in_planes=2, out_planes=1, kernel_size=5, stride=1, padding=2, dilation=1, groups=1, relu=False,
                     bn=False, bias=False)
        bad = {k: v for k, v in cfg.items() if v != built[k] and (v, built[k]) not in (((5, 5), 5), ((2, 2), 2), ((1, 1), 1))}
        if bad:
            _refuse("BasicConv(" + ", ".join(f"{k}={v!r}" for k, v in bad.items()) + "): only BasicConv(2, 1, 5, stride=1, "
                    "padding=2, relu=False) (the 5x5 2->1 conv of CAC_spatial, no BatchNorm, no ReLU, no bias) is built")
        self.out_channels = out_planes
        self.conv = Conv2dParams(in_planes, out_planes, 5)
        self.bn = None
        self.relu = None

    def forward(self, x):
        _refuse("BasicConv.forward: parameter holder; the conv runs inside CODONNet.forward (cac_spatial_kernel)")


class Flatten(nn.Module):  # CAC_module.py:22-24 (index 0 of the mlp Sequential; no parameters)
    def forward(self, x):
        return x.view(x.size(0), -1)


class CAC_channel(nn.Module):
    """CAC_module.py:26-36.  Built: gate_channels=128, reduction_ratio=16 (the 128 -> 8 -> 64 MLP of cac_gate_kernel),
    pool_types = avg + max.  Other arguments are refused at construction (and again at the first forward, for modules that
    arrive by pickle or have their attributes edited)."""

    def __init__(self, gate_channels, reduction_ratio=16, pool_types=("avg", "max")):
        super().__init__()
        if gate_channels != 128 or reduction_ratio != 16:
            _refuse(f"CAC_channel(gate_channels={gate_channels}, reduction_ratio={reduction_ratio}): only (128, 16) is built")
        _check_pool_types("CAC_channel", pool_types)
        self.gate_channels = gate_channels
        self.mlp = nn.Sequential(Flatten(), nn.Linear(gate_channels, gate_channels // reduction_ratio), nn.ReLU(),
                                 nn.Linear(gate_channels // reduction_ratio, gate_channels // 2))
        self.pool_types = list(pool_types)

    def forward(self, x):
        _refuse("CAC_channel.forward: parameter holder; the gate runs inside CODONNet.forward (cac_gate_kernel)")


class ChannelGate(nn.Module):
    """attention/ResCBAM.py:26-37 -- state only in CODONNet (attention_c5, CODON_x4.py:64: registered, never called);
    executed by BaseNet_RMCR_fuseRMCR_cross only, as ChannelGate(64): (64, 16, avg + max) is what is built."""

    def __init__(self, gate_channels, reduction_ratio=16, pool_types=("avg", "max")):
        super().__init__()
        if gate_channels != 64 or reduction_ratio != 16:
            _refuse(f"ChannelGate(gate_channels={gate_channels}, reduction_ratio={reduction_ratio}): only (64, 16) is built")
        _check_pool_types("ChannelGate", pool_types)
        self.gate_channels = gate_channels
        self.mlp = nn.Sequential(Flatten(), nn.Linear(gate_channels, gate_channels // reduction_ratio), nn.ReLU(),
                                 nn.Linear(gate_channels // reduction_ratio, gate_channels))
        self.pool_types = list(pool_types)

    def forward(self, x):
        _refuse("ChannelGate.forward: parameter holder (never called on the CODONNet path, CODON_x4.py:64)")


class ChannelPool(nn.Module):  # CAC_module.py:78-81 (no parameters; fused into cac_stats)
    def forward(self, x):
        _refuse("ChannelPool.forward: no parameters, fused into the statistics kernels of CODONNet.forward")


class CAC_spatial(nn.Module):  # CAC_module.py:83-89
    def __init__(self):
        super().__init__()
        self.compress = ChannelPool()
        self.spatial = BasicConv(2, 1, 5, stride=1, padding=2, relu=False)

    def forward(self, x):
        _refuse("CAC_spatial.forward: parameter holder; the gate runs inside CODONNet.forward (cac_spatial_kernel)")


def _check_gates(model, n_gates: int = 5, gate5: bool = False):
    """First-forward audit of the gate modules as they ARE (a whole-module pickle of the reference bypasses the
    constructors above, and attributes can be edited): what the kernels do not implement raises here."""
    for i in range(n_gates):
        ac, asp = getattr(model, f"attention_c{i}"), getattr(model, f"attention_s{i}")
        _check_pool_types(f"attention_c{i}", ac.pool_types)
        if tuple(ac.mlp[1].weight.shape) != (8, 128) or tuple(ac.mlp[3].weight.shape) != (64, 8):
            _refuse(f"attention_c{i}: MLP {tuple(ac.mlp[1].weight.shape)} -> {tuple(ac.mlp[3].weight.shape)}; only 128 -> 8 -> 64 is built")
        _check_spatial(f"attention_s{i}", asp)
    if gate5:
        g5 = model.attention_c5
        _check_pool_types("attention_c5", g5.pool_types)
        if tuple(g5.mlp[1].weight.shape) != (4, 64) or tuple(g5.mlp[3].weight.shape) != (64, 4):
            _refuse(f"attention_c5: MLP {tuple(g5.mlp[1].weight.shape)} -> {tuple(g5.mlp[3].weight.shape)}; only 64 -> 4 -> 64 is built")
        _check_spatial("attention_s5", model.attention_s5)


def _check_spatial(name: str, asp):
    sc = asp.spatial
    if getattr(sc, "bn", None) is not None or getattr(sc, "relu", None) is not None:
        _refuse(f"{name}.spatial with BatchNorm / ReLU (BasicConv(bn=True / relu=True)): only the bare conv is built")
    conv = sc.conv
    if getattr(conv, "bias", None) is not None or tuple(conv.weight.shape) != (1, 2, 5, 5):
        _refuse(f"{name}.spatial.conv: weight {tuple(conv.weight.shape)}, bias {getattr(conv, 'bias', None) is not None}; "
                "only the bias-free 5x5 2->1 conv is built")
    for attr, want in (("stride", (1, 1)), ("padding", (2, 2)), ("dilation", (1, 1)), ("groups", 1)):
        have = getattr(conv, attr, want)
        if have != want:
            _refuse(f"{name}.spatial.conv.{attr} = {have!r}; only {want!r} is built")


_MAIN_CONVS = [  # (name, cin, cout, k) in the reference's registration order, CODON_x4.py:24-47
    ("input", 1, 64, 3), ("conv_input", 64, 64, 3), ("conv1", 64, 64, 3), ("conv2", 64, 64, 5),
    ("conv3", 128, 128, 5), ("confuse", 128, 64, 1),
    ("input_c", 1, 64, 3), ("conv_input_c", 64, 64, 3), ("conv4", 64, 64, 5), ("conv5", 64, 64, 3),
    ("conv6", 128, 128, 5), ("confuse_c", 128, 64, 1),
    ("conv7", 128, 64, 3), ("conv8", 64, 64, 5), ("conv9", 64, 64, 3), ("conv10", 128, 128, 5),
    ("confuse_fuse", 128, 64, 1), ("conv11", 64, 64, 3), ("output", 64, 1, 3),
]
_MFMA_CONVS = [n for n, ci, co, k in _MAIN_CONVS if ci > 1 and co > 1]


class _CODONBase(nn.Module):
    _HAS_UNUSED_GATE5 = True
    _warned_fp16_eval = False

    def __init__(self):
        super().__init__()
        for name, ci, co, k in _MAIN_CONVS:
            setattr(self, name, Conv2dParams(ci, co, k, he_init=True))
        self.relu = nn.ReLU()
        for i in range(5):
            setattr(self, f"attention_c{i}", CAC_channel(128))
        for i in range(5):
            setattr(self, f"attention_s{i}", CAC_spatial())
        if self._HAS_UNUSED_GATE5:  # CODON_x4.py:64-65: registered, never called
            self.attention_c5 = ChannelGate(64)
            self.attention_s5 = CAC_spatial()
        self._pack_cache: Dict[str, tuple] = {}
        self._wguard: Optional[_WeightGuard] = None
        self.compute_dtype: Optional[torch.dtype] = None
        self.conv_precision: str = "exact"
        self.recompute: bool = False

    def check_supported(self):
        """Raise NotImplementedError if a gate sub-module is configured for something the kernels do not implement
        (pool_types other than avg + max, BatchNorm / ReLU / bias in the spatial conv, other MLP shapes): run by every
        forward on the modules as they are -- a whole-module pickle of the reference bypasses the constructors."""
        _check_gates(self, gate5=isinstance(self, BaseNet_RMCR_fuseRMCR_cross))
        return self

    def set_recompute(self, on: bool = True):
        """Training memory switch: do not keep the 13 `stage` tensors (cat(relu(conv1), relu(conv2)) and siblings,
        CODON_x4.py:79,80,125 -- 128 channels each); the backward re-runs the two sibling convs from the saved block
        input instead.  fp32 at batch 32, 480x640: 232 GB -> 167 GB of the 288 GB, for 26 extra small convs per step."""
        self.recompute = bool(on)
        return self

    def set_conv_precision(self, mode: str):
        """fp32 path only.  "exact" (default): v_mfma_f32_32x32x2_f32, bitwise fp32 fmaf chains.
        "f16x3": OPT-IN split-precision evaluation of the 3x3 / 5x5 convs (operands split into fp16 hi+lo,
        three f16 MFMAs per product, fp32 accumulate; ~2^-22 relative per product, |activations| < 65504) --
        the way off the 157 TF fp32-MFMA ceiling on gfx950, which has no TF32.  Inference only."""
        if mode not in ("exact", "f16x3"):
            raise ValueError(mode)
        self.conv_precision = mode
        return self

    def set_compute_dtype(self, dtype: Optional[torch.dtype]):
        """Activation / MFMA operand dtype: None = follow the parameters' dtype; torch.bfloat16 with fp32
        parameters = bf16 activations and packed weights, fp32 accumulate, fp32 master weights
        (BASELINE.json configs[2], [4]).  Inputs and the output stay fp32 1-channel maps."""
        if dtype not in (None, torch.float32, torch.bfloat16, torch.float16):
            raise NotImplementedError(f"codon_amd.CODONNet: compute dtype {dtype} not supported (fp32, bf16, fp16)")
        self.compute_dtype = dtype
        return self

    def _act_dtype(self) -> torch.dtype:
        dt = self.__dict__.get("compute_dtype")
        if dt is None:
            w = self._modules["input"]._parameters.get("weight")
            dt = (w if w is not None else self._modules["input"].weight).dtype
        if dt not in (torch.float32, torch.bfloat16, torch.float16):
            raise NotImplementedError(f"codon_amd.CODONNet: dtype {dt} not supported (fp32, bf16, fp16)")
        return dt

    # -- packed weights -------------------------------------------------------------------
    def _packed(self, name: str, mode: int = L.PACK_FWD) -> torch.Tensor:
        # (nn.Module.__getattr__ twice per lookup is 1 us, 56 lookups a forward; nn.DataParallel replicas keep their
        # tensors as plain attributes, not in _parameters)
        mod = self._modules[name]
        w = mod._parameters.get("weight")
        if w is None:
            w = mod.weight
        adt = self._act_dtype()
        if mode == L.PACK_FWD and self._split(w.shape[-1]):
            mode = L.PACK_FWD_F16X3
        key = (name, mode, adt)
        tag = (w.data_ptr(), w._version, w.device, w.dtype)
        hit = self._pack_cache.get(key)
        if hit is not None and hit[0] == tag:
            if VERIFY_PACKED and not torch.equal(hit[1], ops.packed_weight(w.detach(), mode, adt)):
                raise RuntimeError(f"codon_amd: the packed image of {name}.weight is stale -- the weight was written "
                                   "through `.data` (or another path that does not bump Tensor._version) after it was "
                                   "packed; call model.invalidate_packed() after such writes")
            return hit[1]
        packed = ops.packed_weight(w.detach(), mode, adt)
        self._pack_cache[key] = (tag, packed)
        return packed

    def invalidate_packed(self):
        """Drop the packed-weight cache.  The cache is keyed on (data_ptr, Tensor._version, device, dtype) of each
        weight; optimizer steps, load_state_dict, .to()/.half()/.cuda() and every other autograd-visible in-place
        op change one of those.  Writes THROUGH `.data` (`w.data.normal_()`, `dist.broadcast(w.data)`) do not bump
        `_version`: after such a write call this method (codon_amd.dist.GradSync does, and load_state_dict / _apply
        are hooked below).  A forgotten call does not go unnoticed: _WeightGuard raises on the next forward."""
        self._pack_cache.clear()
        g = self.__dict__.get("_wguard")
        if g is not None:
            # a forward that ran on stale packed weights and has not been reported yet (the trip is seen one call late) must
            # not be forgotten with the guard's state: wait for the checksum launches enqueued so far and say so
            if g.states and torch.cuda.is_available():
                for d in {st[1].device for st in g.states.values()}:
                    torch.cuda.synchronize(d)
            if g.tripped():
                import warnings
                warnings.warn(_STALE_MSG + " [reported while the packed-weight cache is being invalidated: at least one "
                              "forward BEFORE this point used stale packed weights and its output is wrong]", RuntimeWarning,
                              stacklevel=2)
            g.reset()
        return self

    def _guard(self, dev):
        """One checksum launch per forward over the 17 MFMA conv weights (see _WeightGuard)."""
        if not WEIGHT_GUARD or self.__dict__.get("_no_guard", False):
            return
        g = self.__dict__.get("_wguard")
        if g is None:
            g = self._wguard = _WeightGuard()
        g.run(self, dev)

    def check_packed(self, synchronize: bool = True):
        """Raise if a forward since the last (re)pack ran on stale packed weights (a `.data` write the cache key cannot
        see).  synchronize=True waits for the device first, so every forward enqueued so far has been judged -- the
        natural places: before trusting a result (smoke(), bench warm-up), GraphedCODON.stale()."""
        g = self.__dict__.get("_wguard")
        if g is None:
            return self
        if synchronize and torch.cuda.is_available():
            # every device a checksum launch went to (the module may live on another device than the current one)
            for d in {st[1].device for st in g.states.values()} or {torch.device("cuda", torch.cuda.current_device())}:
                torch.cuda.synchronize(d)
        if g.tripped():
            raise RuntimeError(_STALE_MSG)
        return self

    def _apply(self, fn, *a, **k):
        # weights legitimately replaced: the cache AND a tripped guard go -- after invalidate_packed() has reported a stale
        # forward nobody was told about yet (RuntimeWarning: .half() / .to() / load_state_dict themselves are not at fault)
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super()._load_from_state_dict(*a, **k)

    def _split(self, ksize: int) -> bool:
        return (getattr(self, "conv_precision", "exact") == "f16x3" and ksize in (3, 5)
                and self._act_dtype() == torch.float32)

    def __getstate__(self):  # pickle / deepcopy: drop the device-side cache
        d = self.__dict__.copy()
        d["_pack_cache"] = {}
        d["_wguard"] = None
        d.pop("_grad_sink", None)            # a weakref to the GradSync that owns the gradients: per process, per object
        return d

    def _replicate_for_data_parallel(self):
        r = super()._replicate_for_data_parallel()
        r._pack_cache = {}
        r._wguard = None
        r.__dict__.pop("_grad_sink", None)
        # nn.DataParallel builds fresh replicas (fresh weight copies, empty pack cache) for EVERY forward: nothing can be
        # stale in one, and a guard per call would cost a pinned allocation each time
        r._no_guard = True
        return r

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x, y):  # x: HR-sized depth, y: grey guidance  (CODON_x4.py:66)
        if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 1:
            raise RuntimeError(f"CODONNet expects two (B,1,H,W) tensors, got {tuple(x.shape)} and {tuple(y.shape)}")
        if not x.is_cuda:
            raise RuntimeError("codon_amd.CODONNet runs on MI355X only: move the module and inputs to 'cuda' "
                               "(there is no CPU fallback)")
        adt = self._act_dtype()
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16) or y.dtype != x.dtype:
            raise NotImplementedError(f"codon_amd.CODONNet: input dtype {x.dtype} not supported (fp32, bf16, fp16)")
        if x.shape[0] == 0:
            return self._empty_batch(x, y)
        if torch.is_grad_enabled() and (x.requires_grad or y.requires_grad or
                                        any(p.requires_grad for p in self.parameters())):
            if self.conv_precision != "exact":
                raise NotImplementedError("codon_amd.CODONNet: conv_precision='f16x3' is inference-only "
                                          "(the backward kernels are exact fp32); call under torch.no_grad()")
            if adt == torch.float16:
                # the reference script itself does this: model.cuda().half(), model.eval(), then model(x, y) with
                # grad mode on (test.py:52,66,125).  fp16 has no backward here, so in eval mode the call is served
                # by the inference schedule and returns a detached output; in train mode it is refused.
                if self.training:
                    raise NotImplementedError("codon_amd.CODONNet: fp16 is inference-only (as in the reference, "
                                              "test.py:52); train in fp32 or bf16, or call .eval() first")
                if not _CODONBase._warned_fp16_eval:
                    _CODONBase._warned_fp16_eval = True
                    warnings.warn("codon_amd.CODONNet: fp16 forward in eval mode with grad enabled runs the inference "
                                  "kernels and returns a detached output (no fp16 backward exists)", stacklevel=2)
            else:
                from .autograd import codon_apply  # training path (custom backward)
                return codon_apply(self, x, y)
        # 16-bit inputs (the reference script's `.half()` images, test.py:122-123) go in as they are: _forward_impl converts them
        # in the launch that converts the model's small parameters (codon_cast_multi) -- no ATen cast per image
        # ... and the head stores the output map in the inputs' 16-bit type itself (codon_head_fwd_y16)
        return self._forward_impl(x.contiguous(), y.contiguous(), None, out_dtype=x.dtype)

    def _empty_batch(self, x, y):
        """An empty batch (a rank whose shard of a small global batch holds no image, dist.shard_batch): every op of the
        reference's forward (CODON_x4.py:66-132) accepts it and returns an empty (0,1,H,W) map whose backward leaves ZERO
        gradients in the 44 used parameters (None in attention_c5 / attention_s5).  No kernel has anything to do -- the C ABI
        itself refuses batch 0 -- so the same result is formed here."""
        out = x.new_zeros(x.shape)
        if torch.is_grad_enabled():
            from .autograd import used_parameters
            live = [p for _, p in used_parameters(self) if p.requires_grad] + [t for t in (x, y) if t.requires_grad]
            if live:
                out = out + sum((t.sum() * 0).to(out.dtype) for t in live)
        return out

    def _forward_impl(self, x, y, save: Optional[dict], out_dtype: Optional[torch.dtype] = None):
        """Kernel schedule of CODONNet.forward.  With `save` (a dict) every activation the
        backward needs is kept in fresh buffers; without it buffers are reused across blocks.
        x, y: fp32, or 16-bit (converted here, in the launch that converts a 16-bit model's small parameters);
        out_dtype: dtype of the returned map (default fp32; a 16-bit type only with 16-bit activations of that type)."""
        B, _, H, W = x.shape
        dev = x.device
        self.check_supported()
        self._guard(dev)
        adt = self._act_dtype()
        new = lambda c: ops.new_act(B, c, H, W, adt, dev)
        P = self._packed
        keep = save is not None

        def conv(xs, name, ys, k, **kw):   # one MFMA conv; 3x3 / 5x5 take the split-precision kernel when opted in
            ops.conv2d(xs, P(name), ys, k, f16x3=self._split(k), **kw)

        split5 = self._split(5)
        chain_mode = L.PACK_CHAIN1X1_F16X3 if split5 else L.PACK_CHAIN1X1

        def conv5_1x1(xs, name5, name1, mid, ys, residual=None, stats=None):
            """ys = conv1x1(relu(conv5x5(xs))) [+ residual]; mid = relu(conv5x5(xs)) is only materialised when the
            backward needs it (one launch: the 1x1 runs from the 5x5's accumulators)."""
            ops.conv_chain1x1(xs, P(name5), P(name1, chain_mode), ys, mid=mid if keep else None, residual=residual,
                              f16x3=split5, stats=stats)

        # 16-bit tensors: the CAC statistics of a block come out of the two conv5x5 + 1x1 epilogues (no pass over Fcat).
        # fp32 (round 6): the same for images of at most 32 768 pixels -- chosen by H x W ONLY, and the per-row-strip partials
        # are tiling-invariant, so an image's bits do not depend on the batch it arrives in; larger fp32 images keep the
        # statistics pass (0.87 ms of a 988 ms forward at 32 x 480 x 640, against 5 x 22 us of 2.2 ms at 1 x 128 x 128)
        fused_stats = ops.is_c8(adt) or (FUSED_STATS_F32 and CAC_TAIL and adt == torch.float32 and not split5 and H * W <= 32768)

        # inference, exact fp32: the gate-apply `out*ad_CAC + inputs` (:89-91,117-118) is formed inside the staging of
        # the convs that consume it (codon_conv2d_gated_fwd) instead of a 15 GB HBM pass per block
        # 16-bit: with the emitting conv5x5 (GATED_EMIT) the training forward takes the same route -- the emitted tensor IS
        # the block input the backward needs, bit-identical to cac_apply's output, and the 7.5 GB apply pass is gone there too
        emit16 = GATED_EMIT and not split5 and ((ops.is_c8(adt) and GATED_16BIT) or (adt == torch.float32 and not keep))
        gated = not split5 and (((not keep) and (adt == torch.float32 or GATED_16BIT)) or (keep and emit16))

        emit16 = emit16 and gated
        xg = new(128) if (emit16 and not keep) else None      # [out | out_c] as emitted by the gated conv5x5s of a block

        def gconv(gate, pre_s, in_s, plain_s, name, ys, k, emit=None, emitted=None):
            """relu(conv_k(gate-applied input)): `gate` = (ch, sp) of the producing block or None (plain input).
            emit: this conv also writes the gated input there; emitted: a sibling already did -- run plain on it."""
            if gate is not None and emitted is not None:
                conv(emitted, name, ys, k, relu=True)
            elif gate is not None:
                ops.conv2d_gated(pre_s, in_s, gate[0], gate[1], P(name), ys, k, relu=True, emit=emit)
            else:
                conv(plain_s, name, ys, k, relu=True)

        # the small parameters the kernels take in fp32 (stems, head, the 25 gate tensors): themselves, or -- a model cast to
        # 16 bits as a whole, test.py:52 -- one flat fp32 copy made by ONE launch per forward from the live parameters
        gmods = [(getattr(self, f"attention_c{i}"), getattr(self, f"attention_s{i}")) for i in range(5)]
        # ... and 16-bit input images with them (first in the list: the kernel finds an element's tensor by a linear search)
        small = ops.params_f32([x, y, self.input.weight, self.input_c.weight, self.output.weight] +
                               [t for ac, asp in gmods for t in (ac.mlp[1].weight, ac.mlp[1].bias, ac.mlp[3].weight,
                                                                 ac.mlp[3].bias, asp.spatial.conv.weight)])
        x, y = small[:2]
        w_in, w_in_c, w_out = small[2:5]
        gparams = [small[5 + 5 * i: 10 + 5 * i] for i in range(5)]      # (w1, b1, w2, b2, ws) of block i

        # heads: inputs = in2[:, :64] (depth), inputs_c = in2[:, 64:] (colour)     :68-72
        in2 = new(128)
        t64 = new(64)
        # inference on a small grid (16-bit: at most PAIR_MAX16 tiles; fp32: the small-grid kernels): the depth and the colour
        # conv of every stage as ONE launch
        # (not while bench.py brackets individual conv launches with HIP events: a held launch has no duration of its own)
        pairs = (not keep) and ops.PROFILE is None and \
            B * ((H + 7) // 8) * ((W + 31) // 32) <= (PAIR_MAX16 if ops.is_c8(adt) else min(PAIR_MAX32, 383))
        pair = lambda: ops.conv_pair(dev, pairs)
        t64c = new(64) if (keep or pairs) else t64
        if pairs:
            ops.stem_pair(x, w_in, Slice(t64), y, w_in_c, Slice(t64c))      # both stems as one launch
        else:
            ops.stem(x, w_in, Slice(t64))
        with pair():
            conv(Slice(t64), "conv_input", Slice(in2, 0, 64), 3, relu=True)
            if not pairs:
                ops.stem(y, w_in_c, Slice(t64c))
            conv(Slice(t64c), "conv_input_c", Slice(in2, 64, 64), 3, relu=True)
        inputs, inputs_c = Slice(in2, 0, 64), Slice(in2, 64, 64)
        if keep:
            save["stem"], save["stem_c"], save["in2"] = t64, t64c, in2

        # small grids (inference): two HIP streams, fork before the streams of a block, join at its gate
        two = TWO_STREAMS and (not keep) and (not pairs) and dev.type == "cuda" and (
            B * ((H + 7) // 8) * ((W + 31) // 32) <= TWO_STREAMS_MAX16 if ops.is_c8(adt) else
            B * ((H + 3) // 4) * ((W + 31) // 32) <= TWO_STREAMS_MAX32)
        main_s = torch.cuda.current_stream(dev) if two else None
        halves = _half_chip_streams(dev, main_s) if two else None
        if two:
            # every packed weight image is built (or found) on the CALLER's stream before the fork: an image first built
            # on a side stream would live in that stream's allocator pool and could be recycled there while main-stream
            # kernels still read it
            for n_ in ("conv1", "conv2", "conv4", "conv5", "conv7", "conv8", "conv9", "conv11"):
                P(n_)
            for n5_, n1_ in (("conv3", "confuse"), ("conv6", "confuse_c"), ("conv10", "confuse_fuse")):
                P(n5_)
                P(n1_, chain_mode)

        class _on_half:
            """`with _on_half(k):` runs the body on half-chip stream k, ordered after everything issued so far on the main one."""
            def __init__(self_, k):
                self_.k = k
            def __enter__(self_):
                if two:
                    halves[self_.k].wait_stream(main_s)
                    self_.ctx = torch.cuda.stream(halves[self_.k])
                    self_.ctx.__enter__()
            def __exit__(self_, *a):
                if two:
                    self_.ctx.__exit__(*a)

        def join():
            if two:
                main_s.wait_stream(halves[0])
                main_s.wait_stream(halves[1])

        nt = ops.cac_fused_parts(H, W, adt) if fused_stats else ops.cac_stats_tiles(H, W)
        fz = dict(dtype=torch.float32, device=dev)
        # fp32: the one-launch gate folds the tiles before it finishes the pools -- the serial order of cac_gate_kernel while every
        # fold holds one tile (nt <= 16), and the ONLY sensible form for the many small tiles of a small image (H W <= 32768:
        # 256-pixel tiles, cac.hip), where a serial walk would take longer than the pass itself
        # 16-bit: bit-identical to the separate launches at any size, and FASTER only while the grid is small (12 vs 25 us for
        # one 370 x 463 image; 210 vs 147 us at 32 x 480 x 640, where the combine inside the spatial tiles re-reads four maps'
        # halos) -- so it is chosen by size there; fp32: by H x W only, so that an image's bits never depend on its batch
        tail = CAC_TAIL and ((B * H * W <= CAC_TAIL_MAX_PIXELS or adt == torch.float32) if fused_stats
                             else (nt <= L.CAC_FOLDS or H * W <= 32768))
        if fused_stats:
            pool_c, pool_d = torch.empty((B, 2, H, W), **fz), torch.empty((B, 2, H, W), **fz)
        if fused_stats or tail:
            folded = torch.empty((B, L.CAC_FOLDS, 128, 2), **fz)
        counters = _tail_counters(dev, B) if tail else None    # arrival counters: zero on entry, left at zero by every launch
        cur = in2                       # (B,128): [depth | colour] block input
        oc = prev_gate = prev_pre2 = None
        stage = r2 = stage_c = r2_c = pre2 = None
        for i in range(5):
            drop_stage = keep and getattr(self, "recompute", False)
            if keep or stage is None:
                stage = stage if (drop_stage and stage is not None) else new(128)
                r2, pre2 = new(128), new(128)
                stage_c = (stage_c if (drop_stage and stage_c is not None) else new(128)) if keep else (new(128) if (two or pairs) else stage)
                r2_c = new(128) if (keep or two or pairs) else r2
                pooled = torch.empty((B, 2, H, W), dtype=torch.float32, device=dev)
                partials = torch.empty((B, nt, 128, 2), dtype=torch.float32, device=dev)
                sp = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
            ch = torch.empty((B, 64), dtype=torch.float32, device=dev)
            pools = torch.empty((B, 2, 128), dtype=torch.float32, device=dev) if keep else None
            out, out_c = Slice(cur, 0, 64), Slice(cur, 64, 64)
            pre, pre_c = Slice(pre2, 0, 64), Slice(pre2, 64, 64)
            gate = prev_gate if (gated and i > 0) else None      # (ch, sp) of block i-1: its apply runs in our staging
            # ... on block i-1's [pre | pre_c]: the same buffer at inference, the previous block's saved one in training
            gpre, gpre_c = (Slice(prev_pre2, 0, 64), Slice(prev_pre2, 64, 64)) if gate is not None else (None, None)
            # colour stream: stage_c = [conv4 5x5 | conv5 3x3]                       :76,78,80   (side stream on small grids)
            if emit16 and keep and gate is not None:
                xg = new(128)           # training: the emitted tensor is this block's saved input
            xg_d, xg_c = (Slice(xg, 0, 64), Slice(xg, 64, 64)) if (emit16 and gate is not None) else (None, None)
            if pairs:
                # stage by stage, colour | depth as one launch each (same kernel variant on the same grid)
                with pair():
                    gconv(gate, gpre_c, inputs_c, out_c, "conv4", Slice(stage_c, 0, 64), 5, emit=xg_c)
                    gconv(gate, gpre, inputs, out, "conv2", Slice(stage, 64, 64), 5, emit=xg_d)
                with pair():
                    gconv(gate, gpre_c, inputs_c, out_c, "conv5", Slice(stage_c, 64, 64), 3, emitted=xg_c)
                    gconv(gate, gpre, inputs, out, "conv1", Slice(stage, 0, 64), 3, emitted=xg_d)
                with pair():
                    conv5_1x1(Slice(stage_c), "conv6", "confuse_c", Slice(r2_c), pre_c,
                              stats=(pool_c, partials, 0) if fused_stats else None)
                    conv5_1x1(Slice(stage), "conv3", "confuse", Slice(r2), pre,
                              stats=(pool_d, partials, 64) if fused_stats else None)
            with _on_half(1):
                if not pairs:
                    gconv(gate, gpre_c, inputs_c, out_c, "conv4", Slice(stage_c, 0, 64), 5, emit=xg_c)
                    gconv(gate, gpre_c, inputs_c, out_c, "conv5", Slice(stage_c, 64, 64), 3, emitted=xg_c)
                    conv5_1x1(Slice(stage_c), "conv6", "confuse_c", Slice(r2_c), pre_c,
                              stats=(pool_c, partials, 0) if fused_stats else None)   # :82,83
            # depth stream: stage = [conv1 3x3 | conv2 5x5]                          :75,77,79
            with _on_half(0):
                if not pairs:
                    gconv(gate, gpre, inputs, out, "conv2", Slice(stage, 64, 64), 5, emit=xg_d)
                    gconv(gate, gpre, inputs, out, "conv1", Slice(stage, 0, 64), 3, emitted=xg_d)
                    conv5_1x1(Slice(stage), "conv3", "confuse", Slice(r2), pre,
                              stats=(pool_d, partials, 64) if fused_stats else None)   # :81,84
            join()
            # CAC gate on Fcat = [pre_c | pre]                                       :85-91
            w1_, b1_, w2_, b2_, ws_ = gparams[i]
            if not fused_stats:
                ops.cac_stats(pre_c, pre, pooled, partials)
            if tail:
                # 16-bit inference has no other use for `pooled`: it is formed inside the spatial tiles and not written
                ops.cac_tail(B, H, W, partials, pool_c if fused_stats else None, pool_d if fused_stats else None,
                             pooled if (keep or not fused_stats) else None, folded, counters, w1_, b1_, w2_, b2_, ws_, ch, sp, pools)
            else:
                if fused_stats:
                    ops.cac_fused_finish(B, H, W, partials, pool_c, pool_d, folded, pooled)
                    ops.cac_gate_folded(B, H, W, folded, w1_, b1_, w2_, b2_, ch, pools)
                else:
                    ops.cac_gate(B, H, W, partials, w1_, b1_, w2_, b2_, ch, pools)
                ops.cac_spatial(pooled, ws_, sp)
            if gated:
                prev_gate, prev_pre2 = (ch, sp), pre2    # consumed by the next block's convs / conv7
            else:
                if keep or oc is None:
                    oc = new(128)       # [out | out_c]: also conv7's cat(out, out_c) input  :119
                ops.cac_apply(pre, pre_c, ch, sp, inputs, inputs_c, Slice(oc, 0, 64), Slice(oc, 64, 64))  # :90-91,117-118
            if keep:
                save[f"blk{i}"] = dict(x=xg if (emit16 and gate is not None) else cur, stage=None if drop_stage else stage, r2=r2,
                                       stage_c=None if drop_stage else stage_c, r2_c=r2_c, pre2=pre2,
                                       pooled=pooled, pools=pools, ch=ch, sp=sp)
            if not gated:
                cur = oc

        # fusion trunk                                                               :119-128
        fuse = new(64)
        if gated:
            if keep:
                cur = new(128)          # [out | out_c] of block 4: conv7's input, emitted for the backward
            ops.conv2d_gated(Slice(pre2), Slice(in2), prev_gate[0], prev_gate[1], P("conv7"), Slice(fuse), 3, relu=True,
                             emit=Slice(cur) if keep else None)
        else:
            conv(Slice(cur), "conv7", Slice(fuse), 3, relu=True)
        if keep:
            save["oc"], save["fuse"] = cur, fuse
        f = fuse
        if not keep:
            fA = new(64)
        for i in range(3):
            if keep:
                stage = stage if (drop_stage and stage is not None) else new(128)
                r2, fA = new(128), new(64)
            if pairs:
                # conv8 (5x5) | conv9 (3x3) on the same input: two different kernel bodies as ONE grid (mix53) when the launcher
                # has that form for this dtype and grid, else two launches in order
                with pair():
                    conv(Slice(f), "conv8", Slice(stage, 0, 64), 5, relu=True)    # :123
                    conv(Slice(f), "conv9", Slice(stage, 64, 64), 3, relu=True)   # :124
            else:
                with _on_half(1):
                    conv(Slice(f), "conv9", Slice(stage, 64, 64), 3, relu=True)   # :124
                with _on_half(0):
                    conv(Slice(f), "conv8", Slice(stage, 0, 64), 5, relu=True)    # :123
                join()
            conv5_1x1(Slice(stage), "conv10", "confuse_fuse", Slice(r2), Slice(fA), residual=Slice(fuse))  # :126-128
            if keep:
                save[f"trunk{i}"] = dict(x=f, stage=None if drop_stage else stage, r2=r2)
            f = fA
        # tail                                                                       :129-132
        t = new(64) if keep else t64
        conv(Slice(f), "conv11", Slice(t), 3, relu=True)
        if out_dtype is not None and out_dtype != torch.float32 and out_dtype == adt:
            outp = torch.empty(x.shape, dtype=out_dtype, device=dev)     # the head rounds once, in its store
            ops.head(Slice(t), w_out, x, outp)
        else:
            outp = torch.empty_like(x)
            ops.head(Slice(t), w_out, x, outp)
            if out_dtype is not None and out_dtype != torch.float32:
                outp = outp.to(out_dtype)     # e.g. bf16 inputs to an fp32 model: not the reference's use, one ATen cast
        if keep:
            save["f_last"], save["t11"] = f, t
        return outp


class CODONNet(_CODONBase):
    """x4 / x8 form: 49 state tensors (CODON_X4/CODON_x4.py:18-65)."""
    _HAS_UNUSED_GATE5 = True


class CODONNet16(_CODONBase):
    """x16 form: 44 state tensors, no attention_c5/attention_s5 (CODON_X16/CODON_x16.py:92-135)."""
    _HAS_UNUSED_GATE5 = False


class BaseNet_RMCR_fuseRMCR(nn.Module):
    """Conv-only ablation of the paper (no CAC gates, the two streams never interact before conv7):
    /root/reference/CODON_X16/CODON_x16.py:16-90.  Same 19 bias-free convs, same kernels; inference only."""

    def __init__(self):
        super().__init__()
        for name, ci, co, k in _MAIN_CONVS:
            setattr(self, name, Conv2dParams(ci, co, k, he_init=True))
        self.relu = nn.ReLU()
        self._pack_cache: Dict[str, tuple] = {}
        self._wguard: Optional[_WeightGuard] = None
        self.compute_dtype: Optional[torch.dtype] = None
        self.conv_precision: str = "exact"

    set_compute_dtype = _CODONBase.set_compute_dtype
    _guard = _CODONBase._guard
    check_packed = _CODONBase.check_packed
    set_conv_precision = _CODONBase.set_conv_precision
    _act_dtype = _CODONBase._act_dtype
    _packed = _CODONBase._packed
    _split = _CODONBase._split
    __getstate__ = _CODONBase.__getstate__
    invalidate_packed = _CODONBase.invalidate_packed

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_packed()
        return super()._load_from_state_dict(*a, **k)

    def forward(self, x, y):
        if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 1:
            raise RuntimeError(f"expects two (B,1,H,W) tensors, got {tuple(x.shape)} and {tuple(y.shape)}")
        if not x.is_cuda:
            raise RuntimeError("codon_amd runs on MI355X only (there is no CPU fallback)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("BaseNet_RMCR_fuseRMCR: inference only; call under torch.no_grad()")
        if x.shape[0] == 0:
            return x.new_zeros(x.shape)         # empty batch: an empty map, as the reference's ops return
        idt = x.dtype
        x, y = x.float().contiguous(), y.float().contiguous()
        B, _, H, W = x.shape
        self._guard(x.device)
        adt = self._act_dtype()
        new = lambda c: ops.new_act(B, c, H, W, adt, x.device)
        f32 = lambda t: t if t.dtype == torch.float32 else t.float()
        P = self._packed
        S3, S5 = self._split(3), self._split(5)
        CM = L.PACK_CHAIN1X1_F16X3 if S5 else L.PACK_CHAIN1X1
        t64, stage, oc = new(64), new(128), new(128)

        def stream(img, w_in, n_ci, c3x3, c5x5, first5, n3, nconf, out_slice):      # :53-74
            inputs = new(64)
            ops.stem(img, f32(getattr(self, w_in).weight), Slice(t64))
            ops.conv2d(Slice(t64), P(n_ci), Slice(inputs), 3, relu=True, f16x3=S3)
            cur = Slice(inputs)
            for i in range(5):
                a, b = (c5x5, c3x3) if first5 else (c3x3, c5x5)
                ops.conv2d(cur, P(a), Slice(stage, 0, 64), 5 if first5 else 3, relu=True, f16x3=S5 if first5 else S3)
                ops.conv2d(cur, P(b), Slice(stage, 64, 64), 3 if first5 else 5, relu=True, f16x3=S3 if first5 else S5)
                dst = out_slice if i == 4 else Slice(new(64))
                ops.conv_chain1x1(Slice(stage), P(n3), P(nconf, CM), dst, residual=Slice(inputs), f16x3=S5)  # confuse(relu(conv3)) + inputs
                cur = dst

        stream(x, "input", "conv_input", "conv1", "conv2", False, "conv3", "confuse", Slice(oc, 0, 64))
        stream(y, "input_c", "conv_input_c", "conv5", "conv4", True, "conv6", "confuse_c", Slice(oc, 64, 64))
        fuse, fA = new(64), new(64)
        ops.conv2d(Slice(oc), P("conv7"), Slice(fuse), 3, relu=True, f16x3=S3)                # :76-77
        f = fuse
        for _ in range(3):                                                          # :79-85
            ops.conv2d(Slice(f), P("conv8"), Slice(stage, 0, 64), 5, relu=True, f16x3=S5)
            ops.conv2d(Slice(f), P("conv9"), Slice(stage, 64, 64), 3, relu=True, f16x3=S3)
            ops.conv_chain1x1(Slice(stage), P("conv10"), P("confuse_fuse", CM), Slice(fA), residual=Slice(fuse), f16x3=S5)
            f = fA
        ops.conv2d(Slice(f), P("conv11"), Slice(t64), 3, relu=True, f16x3=S3)                 # :87
        out = torch.empty_like(x)
        ops.head(Slice(t64), f32(self.output.weight), x, out)                       # :88-89
        return out if idt == torch.float32 else out.to(idt)


class BaseNet_RMCR_fuseRMCR_cross(_CODONBase):
    """Sequential-gate ablation: /root/reference/CODON_X4/base_net_withoutBN.py:2186-2317.  Same 49
    state tensors as CODONNet (here attention_c5 / attention_s5 ARE used); inference only.  Differences from CODONNet:
      * the spatial gate of a block is computed on the CHANNEL-GATED features (ops.cac_stats_scaled), :2256-2261;
      * fuse passes through ChannelGate(64) -- whose forward returns x * scale, so the caller's `fuse * gate` squares
        fuse (ops.ew_sq_scale) -- and a spatial gate, with a residual, before the fusion trunk, :2298-2304.
    PARITY UNPINNED: the reference file cannot be imported and takes CHANNEL / SPATIAL from a module it does not ship;
    they are assumed to be CAC_channel / CAC_spatial as in the released CODON_x4.py:5 (oracle.forward_cross, same note)."""
    _HAS_UNUSED_GATE5 = True

    def forward(self, x, y):
        if x.shape != y.shape or x.dim() != 4 or x.shape[1] != 1:
            raise RuntimeError(f"expects two (B,1,H,W) tensors, got {tuple(x.shape)} and {tuple(y.shape)}")
        if not x.is_cuda:
            raise RuntimeError("codon_amd runs on MI355X only (there is no CPU fallback)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise NotImplementedError("BaseNet_RMCR_fuseRMCR_cross: inference only; call under torch.no_grad()")
        if x.shape[0] == 0:
            return x.new_zeros(x.shape)         # empty batch: an empty map, as the reference's ops return
        idt = x.dtype
        x, y = x.float().contiguous(), y.float().contiguous()
        B, _, H, W = x.shape
        dev = x.device
        self.check_supported()
        self._guard(dev)
        adt = self._act_dtype()
        new = lambda c: ops.new_act(B, c, H, W, adt, dev)
        f32 = lambda t: t if t.dtype == torch.float32 else t.float()
        fz = dict(dtype=torch.float32, device=dev)
        P = self._packed
        S3, S5 = self._split(3), self._split(5)
        CM = L.PACK_CHAIN1X1_F16X3 if S5 else L.PACK_CHAIN1X1
        conv = lambda xs, name, ys, k, **kw: ops.conv2d(xs, P(name), ys, k, f16x3=self._split(k), **kw)

        in2, t64, stage, pre2, oc = new(128), new(64), new(128), new(128), new(128)
        ops.stem(x, f32(self.input.weight), Slice(t64))
        conv(Slice(t64), "conv_input", Slice(in2, 0, 64), 3, relu=True)
        ops.stem(y, f32(self.input_c.weight), Slice(t64))
        conv(Slice(t64), "conv_input_c", Slice(in2, 64, 64), 3, relu=True)
        nt = ops.cac_stats_tiles(H, W)
        pooled = torch.empty((B, 2, H, W), **fz)
        partials = torch.empty((B, nt, 128, 2), **fz)
        sp = torch.empty((B, 1, H, W), **fz)
        ch = torch.empty((B, 64), **fz)
        cur = in2
        for i in range(5):
            out, out_c = Slice(cur, 0, 64), Slice(cur, 64, 64)
            pre, pre_c = Slice(pre2, 0, 64), Slice(pre2, 64, 64)
            conv(out, "conv1", Slice(stage, 0, 64), 3, relu=True)
            conv(out, "conv2", Slice(stage, 64, 64), 5, relu=True)
            ops.conv_chain1x1(Slice(stage), P("conv3"), P("confuse", CM), pre, f16x3=S5)
            conv(out_c, "conv4", Slice(stage, 0, 64), 5, relu=True)
            conv(out_c, "conv5", Slice(stage, 64, 64), 3, relu=True)
            ops.conv_chain1x1(Slice(stage), P("conv6"), P("confuse_c", CM), pre_c, f16x3=S5)
            ac, asp = getattr(self, f"attention_c{i}"), getattr(self, f"attention_s{i}")
            ops.cac_stats(pre_c, pre, pooled, partials)
            ops.cac_gate(B, H, W, partials, f32(ac.mlp[1].weight), f32(ac.mlp[1].bias), f32(ac.mlp[3].weight),
                         f32(ac.mlp[3].bias), ch)
            ops.cac_stats_scaled(pre_c, pre, ch, pooled, partials)      # ChannelPool of the channel-gated features
            ops.cac_spatial(pooled, f32(asp.spatial.conv.weight), sp)
            ops.cac_apply(pre, pre_c, ch, sp, Slice(in2, 0, 64), Slice(in2, 64, 64), Slice(oc, 0, 64), Slice(oc, 64, 64))
            cur = oc
        fuse, fuse2, fuse_g, dump = new(64), new(64), new(64), t64
        conv(Slice(cur), "conv7", Slice(fuse), 3, relu=True)
        # ChannelGate(64) = Linear(64,4) / Linear(4,64): run on the 128 -> 8 -> 64 gate kernel with zero-padded weights
        # over the statistics of (fuse | fuse) -- every padded term is an exact zero
        g5 = self.attention_c5
        w1 = torch.zeros((8, 128), **fz); w1[:4, :64] = f32(g5.mlp[1].weight)
        b1 = torch.zeros((8,), **fz); b1[:4] = f32(g5.mlp[1].bias)
        w2 = torch.zeros((64, 8), **fz); w2[:, :4] = f32(g5.mlp[3].weight)
        ops.cac_stats(Slice(fuse), Slice(fuse), pooled, partials)
        ops.cac_gate(B, H, W, partials, w1, b1, w2, f32(g5.mlp[3].bias), ch)
        ops.ew_sq_scale(Slice(fuse), ch, Slice(fuse2))                  # fuse * (fuse * scale)
        ops.cac_stats(Slice(fuse2), Slice(fuse2), pooled, partials)
        ops.cac_spatial(pooled, f32(self.attention_s5.spatial.conv.weight), sp)
        ones = torch.ones((B, 64), **fz)
        ops.cac_apply(Slice(fuse2), Slice(fuse2), ones, sp, Slice(fuse), Slice(fuse), Slice(fuse_g), Slice(dump))
        f, fA = fuse_g, new(64)
        for _ in range(3):
            conv(Slice(f), "conv8", Slice(stage, 0, 64), 5, relu=True)
            conv(Slice(f), "conv9", Slice(stage, 64, 64), 3, relu=True)
            ops.conv_chain1x1(Slice(stage), P("conv10"), P("confuse_fuse", CM), Slice(fA), residual=Slice(fuse_g), f16x3=S5)
            f = fA
        conv(Slice(f), "conv11", Slice(t64), 3, relu=True)
        out = torch.empty_like(x)
        ops.head(Slice(t64), f32(self.output.weight), x, out)
        return out if idt == torch.float32 else out.to(idt)


def strip_module_prefix(state_dict):
    """x16 checkpoints are saved from nn.DataParallel (CODON_X16/test.py:52,60): keys carry 'module.'."""
    return {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
