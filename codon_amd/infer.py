"""python -m codon_amd.infer -- the reference's test loop (CODON_X4/test.py:60-145) on MI355X:
for every image: read depth (already HR-sized) + guidance, forward, clip/*255/uint8, write PNG, masked RMSE
vs the label, SSIM vs the label; print per-image values and the means.  Everything numeric runs in HIP kernels
(codon_amd.CODONNet, codon_amd.metrics); this file is I/O glue."""
from __future__ import annotations

import argparse
import os

import torch

from . import CODONNet, CODONNet16, io, metrics


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--scale", type=int, default=4, choices=[4, 8, 16])
    ap.add_argument("--input-depth", required=True)
    ap.add_argument("--input-color", required=True)
    ap.add_argument("--label", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--weights", default=None, help="X4.pth-style checkpoint; random reference init if absent")
    ap.add_argument("--dtype", default="f16", choices=["f32", "bf16", "f16"], help="the reference script runs .half()")
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("No GPU found, codon_amd has no CPU path")          # test.py:37-38
    dev = torch.device("cuda:0")
    model = (CODONNet16 if a.scale == 16 else CODONNet)()
    if a.weights:
        print("loaded epoch", io.load_checkpoint(a.weights, model))
    else:
        print("WARNING: no --weights given (the reference's X*.pth are not shipped): random init")
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    model = model.to(dev).to(tdt).eval()
    if a.out:
        os.makedirs(a.out, exist_ok=True)
    rm_sum = ss_sum = 0.0
    n = 0
    for f in sorted(os.listdir(a.input_color)):
        dpath = os.path.join(a.input_depth, f)
        if not os.path.exists(dpath):
            continue
        x = io.to_input(io.read_gray(dpath)).to(dev).to(tdt)
        y = io.to_input(io.read_gray(os.path.join(a.input_color, f))).to(dev).to(tdt)
        h, w = min(x.shape[2], y.shape[2]), min(x.shape[3], y.shape[3])
        with torch.no_grad():
            out = model(x[:, :, :h, :w].contiguous(), y[:, :, :h, :w].contiguous())
        out_u8 = metrics.postprocess_u8(out[0, 0])
        if a.out:
            io.write_gray(os.path.join(a.out, f), out_u8.cpu().numpy())
        line = f
        if a.label:
            lab = torch.from_numpy(io.read_gray(os.path.join(a.label, f)).copy()).to(dev)
            rm = metrics.masked_rmse(lab, out_u8)
            ss = metrics.ssim(lab[:h, :w].float() / 255, out_u8.float() / 255)
            rm_sum += rm; ss_sum += ss
            line += f" {rm} {ss}"
        print(line)
        n += 1
    print(n)
    if a.label and n:
        print(rm_sum / n, ss_sum / n)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
