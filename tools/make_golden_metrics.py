#!/usr/bin/env python3
"""Fixtures for the metric kernels: runs only where /root/reference exists.

  * imports the reference's own ssim_2.py (CPU, numpy+scipy) and records ssim_exact on crops of the shipped
    Middlebury sample PNGs (data files of the reference: input_label/, output/, input_depth/);
  * records masked RMSE of the same crops with a literal transcription of test.py::EvaluationResults
    (test.py itself needs cv2 and cannot be imported);
  * records the dataset means of RMSE(output/, input_label/) for x4/x8/x16 (SURVEY.md section 6 quotes
    1.778 / 3.479 / 5.803).
Writes tests/golden/metrics_crops.npz (uint8 crops + float64 expected values).
"""
import os
import sys

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from PIL import Image

from oracle import metrics_oracle as mo

REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "CODON_X4"))
import ssim_2  # noqa: E402  the reference's metric module


def gray(path):
    return np.asarray(Image.open(path).convert("L"), dtype=np.uint8)


def main():
    rec = {}
    crops = [("Art", 40, 60, 96, 128), ("Books", 100, 150, 75, 93), ("Dolls", 0, 0, 64, 64)]
    for name, y0, x0, h, w in crops:
        lab = gray(f"{REF}/CODON_X4/input_label/{name}.png")
        out = gray(f"{REF}/CODON_X4/output/{name}.png")
        dep = gray(f"{REF}/CODON_X4/input_depth/{name}.png")
        lab, dep = lab[:out.shape[0], :out.shape[1]], dep[:out.shape[0], :out.shape[1]]
        sl = (slice(y0, y0 + h), slice(x0, x0 + w))
        for tag, arr in (("label", lab), ("output", out), ("depth", dep)):
            rec[f"{name}.{tag}"] = np.ascontiguousarray(arr[sl])
        rec[f"{name}.ssim_out_label"] = np.float64(ssim_2.ssim_exact(out[sl] / 255, lab[sl] / 255))
        rec[f"{name}.ssim_dep_label"] = np.float64(ssim_2.ssim_exact(dep[sl] / 255, lab[sl] / 255))
        rec[f"{name}.rmse_out_label"] = np.float64(mo.masked_rmse_loop(lab[sl], out[sl]))
        rec[f"{name}.rmse_dep_label"] = np.float64(mo.masked_rmse_loop(lab[sl], dep[sl]))
    for s in (4, 8, 16):
        d = f"{REF}/CODON_X{s}"
        vals = []
        for f in sorted(os.listdir(f"{d}/output")):
            vals.append(mo.masked_rmse(gray(f"{d}/input_label/{f}"), gray(f"{d}/output/{f}")))
        rec[f"dataset_mean_rmse_x{s}"] = np.float64(np.mean(vals))
        print(f"x{s}: mean masked RMSE of shipped outputs = {np.mean(vals):.3f} over {len(vals)} images")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics_crops.npz"), **rec)
    print({k: float(v) for k, v in rec.items() if not hasattr(v, "shape") or v.shape == ()})


def rmcr():
    """Golden outputs of the conv-only ablation class BaseNet_RMCR_fuseRMCR (CODON_X16/CODON_x16.py:16-90)."""
    import torch
    from oracle import codon_oracle as orc
    sys.path.insert(0, os.path.join(REF, "CODON_X16"))
    import CODON_x16 as R
    net = R.BaseNet_RMCR_fuseRMCR().eval()
    sd = {k: torch.from_numpy(orc.kat_tensor(k, tuple(v.shape))) for k, v in net.state_dict().items()}
    net.load_state_dict(sd)
    rec = {}
    for nm, (B, H, W) in (("a", (2, 20, 28)), ("b", (1, 9, 37))):
        x, y = orc.kat_inputs(B, H, W)
        with torch.no_grad():
            rec[f"{nm}.out"] = net(x, y).numpy()
        rec[f"{nm}.shape"] = np.array([B, H, W])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "rmcr_kat0.npz"), **rec)


if __name__ == "__main__":
    main()
    rmcr()
