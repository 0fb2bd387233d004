#!/usr/bin/env python3
"""Numerical prototype (CPU, no kernel): can a 3-way bf16 split of the fp32 operands carry the fp32 headline?

VERDICT r2 item 5: f16x3 (2-way fp16 split, 22 of 24 significand bits, 3 MFMAs per product) is MORE accurate than exact
fp32 MFMA on every He-init fixture and LESS accurate on every KAT-0 fixture, so it may not replace the headline.  A
3-way bf16 split x = b1 + b2 + b3 carries all 24 bits with no range limit; keeping the 6 products of order <= 2
(b1b1, b1b2, b2b1, b1b3, b3b1, b2b2) costs 6 v_mfma_f32_32x32x16_bf16 per 16 channels against 8 v_mfma_f32_32x32x2_f32.

This script emulates the whole network (oracle.forward with its conv replaced) for every golden fixture and compares
each mode's RMSE against the reference's fp64 output (`out_fp64`):
  exact : one fp32 rounding per product-accumulate, k in kernel order (validates the emulation against the GPU table
          profiles/r02_precision_table.txt)
  f16x3 : 2-way fp16 split, 3 terms (validates likewise)
  b3t6 / b3t8 / b3t9 : 3-way bf16 split, 6 / 8 / 9 terms
Model of a 16-bit MFMA: the 16 products of one instruction are summed exactly, the sum is added to the fp32 accumulator
with ONE rounding (the hardware's internal order is not documented; this is the favourable reading).
Run: python tools/bf16x3_study.py [case-substring ...]   (about a minute per small fixture on 8 cores)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.nn.functional as F

from oracle import codon_oracle as orc
from tests.util import GOLDEN_CASES, load_case, rmse

torch.set_num_threads(8)


def split_bf16(t, n):
    parts, r = [], t.float()
    for _ in range(n):
        b = r.bfloat16().float()
        parts.append(b)
        r = r - b                      # exact in fp32
    return parts


def split_f16(t, scale=1.0):
    hi = (t * scale).half().float()
    lo = (t * scale - hi).half().float()
    return [hi, lo]


TERMS = {"b3t6": [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)],
         "b3t8": [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1), (1, 2), (2, 1)],
         "b3t9": [(i, j) for i in range(3) for j in range(3)],
         "f16x3": [(0, 0), (0, 1), (1, 0)]}


def conv_emul(x, w, mode):
    """x (B,Cin,H,W) fp32, w (Cout,Cin,k,k) fp32 -> fp32, arithmetic of `mode`."""
    cout, cin, k, _ = w.shape
    if cin < 16 or cout < 16:          # stem / head: fp32 VALU stencils in the product
        return F.conv2d(x, w, None, 1, k // 2)
    B, _, H, W = x.shape
    p = k // 2
    xp = F.pad(x, (p, p, p, p))
    acc = torch.zeros((B, cout, H, W), dtype=torch.float32)
    if mode == "exact":
        # v_mfma_f32_32x32x2_f32 chains: one rounding per product-accumulate, order (8-channel chunk, dy, dx, channel)
        xd, wd = xp.double(), w.double()
        for c0 in range(0, cin, 8):
            for dy in range(k):
                for dx in range(k):
                    for c in range(c0, c0 + 8):
                        prod = xd[:, c:c + 1, dy:dy + H, dx:dx + W] * wd[None, :, c, dy, dx, None, None]   # exact (48 bits)
                        acc = (acc.double() + prod).float()
        return acc
    if mode == "f16x3":
        xs = [t.double() for t in split_f16(xp)]
        ws = [t.double() for t in split_f16(w, 1024.0)]   # the kernel pre-scales weights by 2^10 (exact), rescales at the end
        post = 1.0 / 1024.0
    else:
        xs = [t.double() for t in split_bf16(xp, 3)]
        ws = [t.double() for t in split_bf16(w, 3)]
        post = 1.0
    for c0 in range(0, cin, 16):
        for dy in range(k):
            for dx in range(k):
                for (i, j) in TERMS[mode]:
                    part = torch.einsum("bchw,oc->bohw", xs[i][:, c0:c0 + 16, dy:dy + H, dx:dx + W], ws[j][:, c0:c0 + 16, dy, dx])
                    acc = (acc.double() + part).float()
    return acc * post if post != 1.0 else acc


def run(sd, x, y, mode):
    old = orc._conv
    orc._conv = lambda a, b: conv_emul(a, b, mode)
    try:
        with torch.no_grad():
            return orc.forward(sd, x, y)
    finally:
        orc._conv = old


def main():
    want = sys.argv[1:]
    modes = os.environ.get("CODON_MODES", "exact,f16x3,b3t6,b3t8").split(",")
    rows = []
    for name in GOLDEN_CASES:
        if want and not any(s in name for s in want):
            continue
        z, variant, sd, x, y = load_case(name)
        if x.numel() > 2 * 32 * 24:
            continue
        ref64 = torch.from_numpy(z["out_fp64"])
        row = {"case": name, "reference_fp32_cpu": rmse(torch.from_numpy(z["out"]), ref64)}
        for m in modes:
            row[m] = rmse(run(sd, x, y, m), ref64)
        rows.append(row)
        print(f"| {name} | " + " | ".join(f"{row[c]:.3e}" for c in ["reference_fp32_cpu"] + modes) + " |", flush=True)
    print("JSON " + json.dumps(rows))


if __name__ == "__main__":
    main()
