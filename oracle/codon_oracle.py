"""CPU oracle for the CODONNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  The product path (codon_amd/) never imports it and has no CPU
fallback: it raises if the HIP library is missing.

This is a functional restatement, in plain PyTorch-CPU ops, of

  * CODONNet.forward            /root/reference/CODON_X4/CODON_x4.py:66-132
    (CODON_X8/CODON_x8.py is byte-identical; the x16 form
     /root/reference/CODON_X16/CODON_x16.py:136-202 is the same math)
  * CAC_channel.forward         /root/reference/CODON_X4/CAC_module.py:38-63
  * ChannelPool / CAC_spatial   /root/reference/CODON_X4/CAC_module.py:78-94

The arithmetic of the reference lives in ATen (nn.Conv2d / F.avg_pool2d / ...),
so the restatement uses the same textbook ops through torch.nn.functional; an
independent plain-C restatement with fp64 accumulation lives in codon_oracle.c.

Pinning: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4).  The oracle is pinned by tests/golden/*.npz, produced by
tools/make_golden.py, which imports the reference's own Python on CPU and
records its outputs, per-stage intermediates and autograd gradients;
tests/test_oracle.py checks this file against those fixtures.
"""
from __future__ import annotations

import zlib
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# state_dict keys in the reference's registration order
# (/root/reference/CODON_X4/CODON_x4.py:24-65)
CONV_SHAPES = [
    ("input.weight", (64, 1, 3, 3)),
    ("conv_input.weight", (64, 64, 3, 3)),
    ("conv1.weight", (64, 64, 3, 3)),
    ("conv2.weight", (64, 64, 5, 5)),
    ("conv3.weight", (128, 128, 5, 5)),
    ("confuse.weight", (64, 128, 1, 1)),
    ("input_c.weight", (64, 1, 3, 3)),
    ("conv_input_c.weight", (64, 64, 3, 3)),
    ("conv4.weight", (64, 64, 5, 5)),
    ("conv5.weight", (64, 64, 3, 3)),
    ("conv6.weight", (128, 128, 5, 5)),
    ("confuse_c.weight", (64, 128, 1, 1)),
    ("conv7.weight", (64, 128, 3, 3)),
    ("conv8.weight", (64, 64, 5, 5)),
    ("conv9.weight", (64, 64, 3, 3)),
    ("conv10.weight", (128, 128, 5, 5)),
    ("confuse_fuse.weight", (64, 128, 1, 1)),
    ("conv11.weight", (64, 64, 3, 3)),
    ("output.weight", (1, 64, 3, 3)),
]


def state_shapes(variant: str = "x4"):
    """Ordered (key, shape) list.  variant 'x4' == 'x8' (49 tensors), 'x16' (44)."""
    out = list(CONV_SHAPES)
    for i in range(5):
        out += [
            (f"attention_c{i}.mlp.1.weight", (8, 128)),
            (f"attention_c{i}.mlp.1.bias", (8,)),
            (f"attention_c{i}.mlp.3.weight", (64, 8)),
            (f"attention_c{i}.mlp.3.bias", (64,)),
        ]
    for i in range(5):
        out += [(f"attention_s{i}.spatial.conv.weight", (1, 2, 5, 5))]
    if variant in ("x4", "x8"):
        # never executed; state only (CODON_x4.py:64-65, attention/ResCBAM.py:26-37)
        out += [
            ("attention_c5.mlp.1.weight", (4, 64)),
            ("attention_c5.mlp.1.bias", (4,)),
            ("attention_c5.mlp.3.weight", (64, 4)),
            ("attention_c5.mlp.3.bias", (64,)),
            ("attention_s5.spatial.conv.weight", (1, 2, 5, 5)),
        ]
    elif variant != "x16":
        raise ValueError(variant)
    return out


def kat_tensor(name: str, shape) -> np.ndarray:
    """KAT-0 deterministic, torch-RNG-independent parameter generator
    (SURVEY.md section 8c): u_i = ((i*2654435761 + crc32(name)) mod 2^32) / 2^32."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64)
    c = np.uint64(zlib.crc32(name.encode()) & 0xFFFFFFFF)
    u = ((i * np.uint64(2654435761) + c) % np.uint64(1 << 32)).astype(np.float64) / float(1 << 32)
    if len(shape) == 4:
        co, _, kh, kw = shape
        v = (u - 0.5) * 2.0 * np.sqrt(3.0) * np.sqrt(2.0 / (kh * kw * co))
    elif len(shape) == 2:
        v = (u - 0.5) * 2.0 / np.sqrt(shape[1])
    else:
        v = (u - 0.5) * 0.2
    return v.astype(np.float32).reshape(shape)


def kat_state(variant: str = "x4") -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(kat_tensor(k, s)) for k, s in state_shapes(variant)}


def kat_inputs(B: int, H: int, W: int):
    """KAT-0 inputs: x=((37i+101j+13b) mod 256)/255, y=((53i+29j+7b+91) mod 256)/255."""
    b = np.arange(B).reshape(B, 1, 1, 1)
    i = np.arange(H).reshape(1, 1, H, 1)
    j = np.arange(W).reshape(1, 1, 1, W)
    x = ((37 * i + 101 * j + 13 * b) % 256) / 255.0
    y = ((53 * i + 29 * j + 7 * b + 91) % 256) / 255.0
    return torch.from_numpy(x.astype(np.float32)), torch.from_numpy(y.astype(np.float32))


def he_state(variant: str = "x4", seed: int = 0) -> Dict[str, torch.Tensor]:
    """He-normal conv weights (CODON_x4.py:50-53 rule) + uniform CAC params drawn from
    a numpy Generator, so the GPU box rebuilds them without torch RNG."""
    g = np.random.default_rng(seed)
    sd = {}
    for k, s in state_shapes(variant):
        if k.startswith("attention"):
            fan_in = int(np.prod(s[1:])) if len(s) > 1 else 8
            bound = 1.0 / np.sqrt(fan_in)
            sd[k] = torch.from_numpy(g.uniform(-bound, bound, size=s).astype(np.float32))
        else:
            co, _, kh, kw = s
            sd[k] = torch.from_numpy((g.standard_normal(size=s) * np.sqrt(2.0 / (kh * kw * co))).astype(np.float32))
    return sd


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------

def _conv(x, w):
    return F.conv2d(x, w, None, 1, w.shape[-1] // 2)


def cac_channel(Fcat, w1, b1, w2, b2):
    """CAC_channel.forward, CAC_module.py:38-63.  Returns the (B,64) gate (the reference
    expands it to (B,64,H,W); the expansion carries no arithmetic)."""
    H, W = Fcat.shape[2], Fcat.shape[3]
    avg = F.avg_pool2d(Fcat, (H, W), stride=(H, W)).flatten(1)   # :43
    mx = F.max_pool2d(Fcat, (H, W), stride=(H, W)).flatten(1)    # :47

    def mlp(v):                                                  # :30-35
        return F.linear(F.relu(F.linear(v, w1, b1)), w2, b2)

    return torch.sigmoid(mlp(avg) + mlp(mx))                     # :58-62


def cac_spatial(Fcat, ws):
    """CAC_spatial.forward, CAC_module.py:90-94 with ChannelPool :78-81 (max FIRST, mean second)."""
    comp = torch.cat((Fcat.max(1, keepdim=True)[0], Fcat.mean(1, keepdim=True)), 1)
    return torch.sigmoid(F.conv2d(comp, ws, None, 1, 2))


def forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, y: torch.Tensor,
            taps: Optional[dict] = None, masks=None) -> torch.Tensor:
    """CODONNet.forward(x, y), CODON_x4.py:66-132.  `taps`, if given, receives
    per-stage intermediates under stable names.  `masks` (test infrastructure): an iterable of
    boolean tensors, one per ReLU in call order -- each ReLU then is z * mask instead of
    max(z, 0), so two implementations whose pre-activations differ by summation noise around
    zero can be compared on IDENTICAL masks (the gradient is discontinuous there)."""
    if masks is None:
        r = F.relu
    else:
        it = iter(masks)
        r = lambda z: z * next(it).to(z.dtype)
    w = lambda k: sd[k + ".weight"]
    residual = x                                                  # :67
    inputs = r(_conv(r(_conv(x, w("input"))), w("conv_input")))          # :68-69
    inputs_c = r(_conv(r(_conv(y, w("input_c"))), w("conv_input_c")))    # :71-72
    out, out_c = inputs, inputs_c
    if taps is not None:
        taps["inputs"], taps["inputs_c"] = inputs, inputs_c
    for i in range(5):
        R1 = r(_conv(out, w("conv1")))                            # :75  3x3
        P1_c = r(_conv(out_c, w("conv5")))                        # :76  3x3
        P1 = r(_conv(out, w("conv2")))                            # :77  5x5
        R1_c = r(_conv(out_c, w("conv4")))                        # :78  5x5
        stage = torch.cat((R1, P1), 1)                            # :79  3x3 | 5x5
        stage_c = torch.cat((R1_c, P1_c), 1)                      # :80  5x5 | 3x3
        R2 = r(_conv(stage, w("conv3")))                          # :81
        R2_c = r(_conv(stage_c, w("conv6")))                      # :82
        if taps is not None:
            taps[f"blk{i}.stage"], taps[f"blk{i}.stage_c"] = stage, stage_c
            taps[f"blk{i}.r2"], taps[f"blk{i}.r2_c"] = R2, R2_c
        out_c = _conv(R2_c, w("confuse_c"))                       # :83
        out = _conv(R2, w("confuse"))                             # :84
        Fcat = torch.cat((out_c, out), 1)                         # :85  colour | depth
        ch = cac_channel(Fcat, sd[f"attention_c{i}.mlp.1.weight"], sd[f"attention_c{i}.mlp.1.bias"],
                         sd[f"attention_c{i}.mlp.3.weight"], sd[f"attention_c{i}.mlp.3.bias"])
        sp = cac_spatial(Fcat, sd[f"attention_s{i}.spatial.conv.weight"])
        g = ch[:, :, None, None] * sp                             # :89
        if taps is not None:
            taps[f"blk{i}.pre"], taps[f"blk{i}.pre_c"] = out, out_c
            taps[f"blk{i}.ch"], taps[f"blk{i}.sp"] = ch, sp
        out = out * g + inputs                                    # :90, :118
        out_c = out_c * g + inputs_c                              # :91, :117
        if taps is not None:
            taps[f"blk{i}.out"], taps[f"blk{i}.out_c"] = out, out_c
    fuse = r(_conv(torch.cat((out, out_c), 1), w("conv7")))       # :119-120  depth | colour
    f = fuse
    if taps is not None:
        taps["fuse"] = fuse
    for i in range(3):
        st = torch.cat((r(_conv(f, w("conv8"))), r(_conv(f, w("conv9")))), 1)   # :123-125
        r2 = r(_conv(st, w("conv10")))                                          # :126
        f = _conv(r2, w("confuse_fuse")) + fuse                                 # :127-128
        if taps is not None:
            taps[f"trunk{i}"], taps[f"trunk{i}.stage"], taps[f"trunk{i}.r2"] = f, st, r2
    out = r(_conv(f, w("conv11")))                                # :129
    if taps is not None:
        taps["t11"] = out
    return _conv(out, w("output")) + residual                     # :130-132


def forward_rmcr(sd, x, y):
    """BaseNet_RMCR_fuseRMCR.forward, /root/reference/CODON_X16/CODON_x16.py:51-90 (conv-only ablation)."""
    r = F.relu
    w = lambda k: sd[k + ".weight"]

    def stream(img, n_in, n_ci, a, b, n3, nconf):
        inputs = r(_conv(r(_conv(img, w(n_in))), w(n_ci)))
        out = inputs
        for _ in range(5):
            st = torch.cat((r(_conv(out, w(a))), r(_conv(out, w(b)))), 1)
            out = _conv(r(_conv(st, w(n3))), w(nconf)) + inputs
        return out

    out = stream(x, "input", "conv_input", "conv1", "conv2", "conv3", "confuse")          # :53-62
    out_c = stream(y, "input_c", "conv_input_c", "conv4", "conv5", "conv6", "confuse_c")  # :64-74
    fuse = r(_conv(torch.cat((out, out_c), 1), w("conv7")))
    f = fuse
    for _ in range(3):
        st = torch.cat((r(_conv(f, w("conv8"))), r(_conv(f, w("conv9")))), 1)
        f = _conv(r(_conv(st, w("conv10"))), w("confuse_fuse")) + fuse
    return _conv(r(_conv(f, w("conv11"))), w("output")) + x


def forward_cross(sd, x, y):
    """BaseNet_RMCR_fuseRMCR_cross.forward, /root/reference/CODON_X4/base_net_withoutBN.py:2234-2317 -- the sequential
    gate ablation: channel gate first, spatial gate computed on the channel-gated features, and a gated fusion trunk.

    PARITY UNPINNED, twice over: the file is un-importable, and it takes CHANNEL / SPATIAL from a module
    `wechat_guide` that the reference does not ship (:16-17).  This restatement ASSUMES they are the classes the released
    CODON_x4.py imports under the same aliases (`from CAC_module import CAC_channel as CHANNEL, CAC_spatial as SPATIAL`,
    CODON_x4.py:5): both return the gate only.  attention_c5 is attention/ResCBAM.py's ChannelGate, whose forward returns
    x * scale (:60-61) -- so `fuse * attention_c_fuse` (:2298) squares fuse; attention_s5 is SPATIAL (gate only)."""
    r = F.relu
    w = lambda k: sd[k + ".weight"]
    inputs = r(_conv(r(_conv(x, w("input"))), w("conv_input")))          # :2236-2237
    inputs_c = r(_conv(r(_conv(y, w("input_c"))), w("conv_input_c")))    # :2239-2240
    out, out_c = inputs, inputs_c
    for i in range(5):
        stage = torch.cat((r(_conv(out, w("conv1"))), r(_conv(out, w("conv2")))), 1)          # :2243,2245,2247
        stage_c = torch.cat((r(_conv(out_c, w("conv4"))), r(_conv(out_c, w("conv5")))), 1)    # :2244,2246,2248
        out_c = _conv(r(_conv(stage_c, w("conv6"))), w("confuse_c"))     # :2250-2251
        out = _conv(r(_conv(stage, w("conv3"))), w("confuse"))           # :2249,2252
        att_cat = torch.cat((out_c, out), 1)                             # :2253
        ch = cac_channel(att_cat, sd[f"attention_c{i}.mlp.1.weight"], sd[f"attention_c{i}.mlp.1.bias"],
                         sd[f"attention_c{i}.mlp.3.weight"], sd[f"attention_c{i}.mlp.3.bias"])[:, :, None, None]
        out_c, out = out_c * ch, out * ch                                # :2256-2257
        sp = cac_spatial(torch.cat((out_c, out), 1), sd[f"attention_s{i}.spatial.conv.weight"])   # :2258-2259
        out_c = out_c * sp + inputs_c                                    # :2260, :2295
        out = out * sp + inputs                                          # :2261, :2296
    fuse = r(_conv(torch.cat((out, out_c), 1), w("conv7")))              # :2298-2299
    res_fuse = fuse
    H, W = fuse.shape[2], fuse.shape[3]
    avg = F.avg_pool2d(fuse, (H, W), stride=(H, W)).flatten(1)
    mx = F.max_pool2d(fuse, (H, W), stride=(H, W)).flatten(1)
    mlp5 = lambda v: F.linear(F.relu(F.linear(v, sd["attention_c5.mlp.1.weight"], sd["attention_c5.mlp.1.bias"])),
                              sd["attention_c5.mlp.3.weight"], sd["attention_c5.mlp.3.bias"])
    att_c = fuse * torch.sigmoid(mlp5(avg) + mlp5(mx))[:, :, None, None]   # :2301  (ChannelGate returns x * scale)
    fuse = fuse * att_c                                                  # :2302
    fuse = fuse * cac_spatial(fuse, sd["attention_s5.spatial.conv.weight"]) + res_fuse   # :2303-2304
    f = fuse
    for _ in range(3):                                                   # :2306-2312
        st = torch.cat((r(_conv(f, w("conv8"))), r(_conv(f, w("conv9")))), 1)
        f = _conv(r(_conv(st, w("conv10"))), w("confuse_fuse")) + fuse
    return _conv(r(_conv(f, w("conv11"))), w("output")) + x             # :2313-2315


def forward_numpy(sd, x, y):
    with torch.no_grad():
        return forward(sd, torch.as_tensor(x), torch.as_tensor(y)).numpy()


def loss_l1(pred, target):
    """Per-batch mean L1; used as the scalar for gradient fixtures."""
    return (pred - target).abs().mean()


def grads(sd, x, y, target, masks=None, upstream=None):
    """Autograd gradients of loss_l1 w.r.t. every used parameter (44 tensors).  `masks`: see forward();
    `upstream`: use this dL/d(out) instead of the L1 loss's own (its sign is discontinuous too)."""
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    out = forward(p, x, y, masks=masks)
    if upstream is not None:
        used = [k for k in p if not (k.startswith("attention_c5") or k.startswith("attention_s5"))]
        gs = torch.autograd.grad(out, [p[k] for k in used], grad_outputs=upstream)
        return float(loss_l1(out.detach(), target)), {k: g for k, g in zip(used, gs)}, out.detach()
    loss = loss_l1(out, target)
    used = [k for k in p if not (k.startswith("attention_c5") or k.startswith("attention_s5"))]
    gs = torch.autograd.grad(loss, [p[k] for k in used])
    return float(loss.detach()), {k: g for k, g in zip(used, gs)}, out.detach()
