#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REFERENCE's own Python on CPU.

Runs only in the build container (needs /root/reference); the fixtures it writes are
data (inputs are regenerated from formulas, expected outputs are stored) and travel
to the GPU box; the reference itself never does.

  python tools/make_golden.py            # writes tests/golden/

What is recorded, per case:
  out            final CODONNet(x, y)                     (reference forward, fp32 CPU)
  taps           per-stage intermediates captured with forward hooks on the reference
                 modules (conv_input, confuse, confuse_c, attention_c{i}, attention_s{i},
                 conv1 inputs = block outputs, conv7, confuse_fuse)
  grads          autograd gradients of mean|out - target| w.r.t. the 44 used parameters
                 (subsampled: every `stride`-th element + L2 norm + sum)
Weights come from oracle.codon_oracle.kat_state / he_state (numpy, no torch RNG).
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import codon_oracle as orc

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def load_reference(variant):
    d = {"x4": "CODON_X4", "x8": "CODON_X8", "x16": "CODON_X16"}[variant]
    mod = {"x4": "CODON_x4", "x8": "CODON_x8", "x16": "CODON_x16"}[variant]
    path = os.path.join(REF, d)
    for m in ("CAC_module", "attention", "attention.ResCBAM", mod):
        sys.modules.pop(m, None)
    sys.path.insert(0, path)
    try:
        import importlib
        m = importlib.import_module(mod)
        net = m.CODONNet()
    finally:
        sys.path.remove(path)
    return net


def run_reference(net, sd, x, y, want_taps):
    net.load_state_dict(sd, strict=True)
    net.eval()
    taps = {}
    hooks = []
    counters = {}

    def rec(name_fn):
        def h(mod, inp, out):
            k = counters.get(id(mod), 0)
            counters[id(mod)] = k + 1
            nm = name_fn(k)
            if nm:
                taps[nm] = out.detach().clone()
        return h

    if want_taps:
        hooks.append(net.conv_input.register_forward_hook(rec(lambda k: "inputs.prerelu")))
        hooks.append(net.conv_input_c.register_forward_hook(rec(lambda k: "inputs_c.prerelu")))
        hooks.append(net.confuse.register_forward_hook(rec(lambda k: f"blk{k}.pre")))
        hooks.append(net.confuse_c.register_forward_hook(rec(lambda k: f"blk{k}.pre_c")))
        for i in range(5):
            hooks.append(getattr(net, f"attention_c{i}").register_forward_hook(
                (lambda i: lambda m, a, o: taps.__setitem__(f"blk{i}.ch", o[:, :, 0, 0].detach().clone()))(i)))
            hooks.append(getattr(net, f"attention_s{i}").register_forward_hook(
                (lambda i: lambda m, a, o: taps.__setitem__(f"blk{i}.sp", o.detach().clone()))(i)))
        cnt = {"c1": 0, "c4": 0}

        def pre1(mod, inp):
            k = cnt["c1"]; cnt["c1"] += 1
            if k >= 1:
                taps[f"blk{k-1}.out"] = inp[0].detach().clone()

        def pre4(mod, inp):
            k = cnt["c4"]; cnt["c4"] += 1
            if k >= 1:
                taps[f"blk{k-1}.out_c"] = inp[0].detach().clone()

        def pre7(mod, inp):
            taps["blk4.out"] = inp[0][:, :64].detach().clone()
            taps["blk4.out_c"] = inp[0][:, 64:].detach().clone()

        hooks.append(net.conv1.register_forward_pre_hook(pre1))
        hooks.append(net.conv4.register_forward_pre_hook(pre4))
        hooks.append(net.conv7.register_forward_pre_hook(pre7))
        hooks.append(net.conv7.register_forward_hook(rec(lambda k: "fuse.prerelu")))
        hooks.append(net.confuse_fuse.register_forward_hook(rec(lambda k: f"trunk{k}.preadd")))
    with torch.no_grad():
        out = net(x, y)
    for h in hooks:
        h.remove()
    return out, taps


def ref_grads(net, sd, x, y, target):
    net.load_state_dict(sd, strict=True)
    net.zero_grad()
    for p in net.parameters():
        p.requires_grad_(True)
    out = net(x, y)
    loss = (out - target).abs().mean()
    loss.backward()
    gs = {}
    for k, p in net.named_parameters():
        if p.grad is not None:
            gs[k] = p.grad.detach().clone()
    return float(loss), gs


def target_for(x):
    # deterministic smooth "ground truth": not in the reference (it has no loss); just a fixed tensor
    B, _, H, W = x.shape
    i = np.arange(H).reshape(1, 1, H, 1)
    j = np.arange(W).reshape(1, 1, 1, W)
    b = np.arange(B).reshape(B, 1, 1, 1)
    t = 0.5 + 0.45 * np.sin(0.37 * i + 0.11 * b) * np.cos(0.23 * j)
    return torch.from_numpy(t.astype(np.float32))


def sub(g, maxn=4096):
    f = g.flatten()
    stride = max(1, (f.numel() + maxn - 1) // maxn)
    return stride, f[::stride].numpy().copy()


CASES = [
    # name, variant, weights, (B,H,W), taps?, grads?
    ("kat0_x4_2x32x24", "x4", "kat", (2, 32, 24), False, True),
    ("kat0_x4_1x13x11_taps", "x4", "kat", (1, 13, 11), True, False),
    ("he0_x4_2x24x20_taps", "x4", "he", (2, 24, 20), True, True),
    ("he1_x4_1x17x19", "x4", "he1", (1, 17, 19), False, False),
    ("kat0_x4_1x5x5", "x4", "kat", (1, 5, 5), False, False),
    ("kat0_x4_1x1x1", "x4", "kat", (1, 1, 1), False, False),
    ("kat0_x4_1x3x70", "x4", "kat", (1, 3, 70), False, False),
    ("kat0_x8_1x16x16", "x8", "kat", (1, 16, 16), False, False),
    ("kat0_x16_2x20x28", "x16", "kat", (2, 20, 28), False, True),
    ("he0_x16_1x33x9", "x16", "he", (1, 33, 9), False, False),
]


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    nets = {}
    for name, variant, wkind, (B, H, W), want_taps, want_grads in CASES:
        if variant not in nets:
            nets[variant] = load_reference(variant)
        net = nets[variant]
        if wkind == "kat":
            sd = orc.kat_state(variant)
        else:
            sd = orc.he_state(variant, seed={"he": 0, "he1": 1}[wkind])
        # key order / shapes must be the reference's
        ref_sd = net.state_dict()
        assert list(ref_sd.keys()) == list(sd.keys()), "state_dict key order mismatch"
        for k in sd:
            assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), k
        x, y = orc.kat_inputs(B, H, W)
        out, taps = run_reference(net, sd, x, y, want_taps)
        rec = {"out": out.numpy(), "shape": np.array([B, H, W]), "variant": variant, "weights": wkind}
        for k, v in taps.items():
            # the large he0 case keeps a subset of taps so the fixture stays ~1 MB
            if B * H * W > 400 and not (k.endswith(".ch") or k.endswith(".sp") or k in
                                        ("blk0.pre", "blk4.out", "blk4.out_c", "trunk2.preadd")):
                continue
            rec["tap." + k] = v.numpy()
        if want_grads:
            tgt = target_for(x)
            loss, gs = ref_grads(net, sd, x, y, tgt)
            rec["loss"] = np.float64(loss)
            for k, g in gs.items():
                stride, s = sub(g)
                rec["grad." + k] = s
                rec["gradstride." + k] = np.int64(stride)
                rec["gradnorm." + k] = np.float64(g.double().norm())
                rec["gradsum." + k] = np.float64(g.double().sum())
        # fp64 run of the same reference net: bounds the fp32 noise floor
        net64 = net.double()
        with torch.no_grad():
            o64 = net64(x.double(), y.double())
        rec["out_fp64"] = o64.numpy()
        net.float()
        path = os.path.join(GOLD, name + ".npz")
        np.savez_compressed(path, **rec)
        rm = float((out.double() - o64).pow(2).mean().sqrt())
        print(f"{name}: out sum {float(out.double().sum()):.6f} std {float(out.std()) if out.numel()>1 else 0:.6f} "
              f"fp32-vs-fp64 rmse {rm:.3e}  -> {os.path.getsize(path)/1024:.0f} KiB")
    # state_dict key order + shapes (the drop-in contract)
    with open(os.path.join(GOLD, "state_dict_keys.txt"), "w") as f:
        for variant in ("x4", "x8", "x16"):
            if variant not in nets:
                nets[variant] = load_reference(variant)
            for k, v in nets[variant].state_dict().items():
                f.write(f"{variant} {k} {'x'.join(str(d) for d in v.shape)}\n")


if __name__ == "__main__":
    main()
