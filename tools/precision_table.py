#!/usr/bin/env python3
"""Error of each fp32-path conv mode against the reference's own fp64 run (tests/golden/*.npz: out_fp64), for every
network fixture + the config-0 size (1x128x128, fp64 oracle computed here on the CPU).  Settles whether an opt-in
mode is "no less accurate than exact fp32" by data.  Prints a markdown table + JSON."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import codon_oracle as orc
from tests.util import GOLDEN_CASES, load_case, rmse

MODES = [m for m in os.environ.get("CODON_MODES", "exact,f16x3").split(",") if m]


def model_for(variant, sd):
    from codon_amd import CODONNet, CODONNet16
    m = (CODONNet16 if variant == "x16" else CODONNet)()
    m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def fp64_oracle(sd, x, y):
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        return orc.forward(sd64, x.double(), y.double())


def main():
    rows = []
    cases = [(n,) + load_case(n) for n in GOLDEN_CASES]
    extra = []
    for seed, (B, H, W) in ((3, (1, 128, 128)), (4, (2, 96, 160)), (5, (1, 240, 320))):
        sd = orc.he_state("x4", seed=100 + seed)
        g = np.random.default_rng(seed)
        x = torch.from_numpy(g.random((B, 1, H, W), dtype=np.float32))
        y = torch.from_numpy((g.integers(0, 256, size=(B, 1, H, W)) / 255.0).astype(np.float32))
        extra.append((f"he{100 + seed}_x4_{B}x{H}x{W} (fp64 oracle)", sd, x, y, fp64_oracle(sd, x, y)))
    for name, z, variant, sd, x, y in cases:
        extra.append((name, sd, x, y, torch.from_numpy(z["out_fp64"])))
        extra[-1] = extra[-1] + (variant, torch.from_numpy(z["out"]))
    for item in extra:
        name, sd, x, y, ref64 = item[:5]
        variant = item[5] if len(item) > 5 else "x4"
        m = model_for(variant, sd)
        row = {"case": name, "out_std": float(ref64.std()) if ref64.numel() > 1 else 0.0}
        if len(item) > 6:
            row["reference_fp32_cpu"] = rmse(item[6], ref64)
        for mode in MODES:
            m.set_conv_precision(mode)
            with torch.no_grad():
                o = m(x.cuda(), y.cuda()).cpu()
            row[mode] = rmse(o, ref64)
        rows.append(row)
    cols = ["reference_fp32_cpu"] + MODES
    print("| case | out std | " + " | ".join(cols) + " |")
    print("|---|---|" + "---|" * len(cols))
    for r in rows:
        print(f"| {r['case']} | {r['out_std']:.3g} | " + " | ".join(f"{r[c]:.3e}" if c in r else "-" for c in cols) + " |")
    for mode in MODES[1:]:
        wins = sum(1 for r in rows if r[mode] <= r["exact"])
        print(f"{mode}: no less accurate than exact fp32 on {wins} of {len(rows)} cases; "
              f"geomean error ratio {np.exp(np.mean([np.log(max(r[mode], 1e-30) / max(r['exact'], 1e-30)) for r in rows])):.3f}")
    print("JSON " + json.dumps(rows))


if __name__ == "__main__":
    main()
