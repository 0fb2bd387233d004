#!/usr/bin/env python3
"""Round-4 addition to tests/golden/ (same rules as tools/make_golden.py: imports the REFERENCE's own Python on CPU, runs
only in the build container, writes data only; earlier fixtures are left untouched).

  bf16grad_*   the reference MODULE cast to bfloat16 (net.bfloat16(), what /root/reference/CODON_X4/test.py:52 does with
               .half()) run forward AND backward on CPU -- the reference's own bf16 autograd through
               CODON_x4.py:66-132 / CAC_module.py:38-94 -- beside the float64 twin of the same net on the same inputs.
               Both backward passes start from the SAME upstream gradient dL/d(out) = sign(out_fp64 - target) / N (the
               L1 loss's gradient on the fp64 output: the sign is a discontinuity, fixing it isolates the backward pass).
               Per used parameter tensor (44): every `stride`-th element of the fp64 gradient and of the reference-bf16
               gradient, their full L2 norms, and the reference-bf16 error vs fp64 on the full tensor and on the stored
               subsample.  tests/test_gpu_backward.py holds the HIP bf16 path to <= RATIO x that error, tensor by tensor.
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

import make_golden as mg
from oracle import codon_oracle as orc

SEEDS = {"he": 0, "he1": 1, "he2": 2}
CASES = [
    ("bf16grad_he0_x4_2x24x20", "x4", "he", (2, 24, 20)),
    ("bf16grad_he1_x16_1x40x56", "x16", "he1", (1, 40, 56)),
]
MAXN = 8192


def grads_of(net, x, y, up):
    net.zero_grad()
    for p in net.parameters():
        p.requires_grad_(True)
    out = net(x, y)
    out.backward(up.to(out.dtype))
    gs = {k: p.grad.detach().double().clone() for k, p in net.named_parameters() if p.grad is not None}
    for p in net.parameters():
        p.requires_grad_(False)
        p.grad = None
    return out.detach(), gs


def main():
    torch.set_num_threads(8)
    for name, variant, wkind, (B, H, W) in CASES:
        net = mg.load_reference(variant)
        sd = orc.he_state(variant, seed=SEEDS[wkind])
        x, y = orc.kat_inputs(B, H, W)
        tgt = mg.target_for(x)
        net.load_state_dict(sd, strict=True)
        net.train()
        with torch.no_grad():
            o64 = net.double()(x.double(), y.double())
        up = (torch.sign(o64 - tgt.double()) / o64.numel()).float()
        _, g64 = grads_of(net, x.double(), y.double(), up.double())
        ob, gb = grads_of(net.bfloat16(), x.bfloat16(), y.bfloat16(), up)
        net.float()
        rec = {"shape": np.array([B, H, W]), "variant": variant, "weights": wkind, "upstream": up.numpy(),
               "out_fp64": o64.numpy(), "out_bf16": ob.float().numpy()}
        assert len(g64) == 44 and set(g64) == set(gb)
        worst = (0.0, None)
        for k in g64:
            stride, s64 = mg.sub(g64[k], MAXN)
            _, sb = mg.sub(gb[k], MAXN)
            rec["g64." + k] = s64.astype(np.float32)
            rec["gbf16." + k] = sb.astype(np.float32)
            rec["stride." + k] = np.int64(stride)
            rec["norm64." + k] = np.float64(g64[k].norm())
            rec["normbf16." + k] = np.float64(gb[k].norm())
            e_full = float((gb[k] - g64[k]).norm() / g64[k].norm())
            e_sub = float(np.linalg.norm(sb - s64) / np.linalg.norm(s64))
            rec["err_full." + k] = np.float64(e_full)
            rec["err_sub." + k] = np.float64(e_sub)
            worst = max(worst, (e_full, k))
            print(f"  {k:40s} n {g64[k].numel():7d} ref-bf16 vs fp64: full {e_full:.3e} subsample {e_sub:.3e}")
        path = os.path.join(mg.GOLD, name + ".npz")
        np.savez_compressed(path, **rec)
        eo = float((ob.double() - o64).pow(2).mean().sqrt() / o64.pow(2).mean().sqrt())
        print(f"{name}: output bf16-vs-fp64 rel-RMSE {eo:.3e}; worst gradient tensor {worst[1]} {worst[0]:.3e} "
              f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
