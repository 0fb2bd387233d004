"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import os

import numpy as np
import torch

from oracle import codon_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# network fixtures (tools/make_golden.py); metrics_crops.npz and rmcr_kat0.npz have their own tests
GOLDEN_CASES = sorted(f[:-4] for f in os.listdir(GOLD) if f.endswith(".npz") and f.split("_")[0] in ("kat0", "he0", "he1", "he2"))
# the reference MODULE run in bfloat16 on CPU (tools/make_golden_r2.py): pins the bf16 tolerance
BF16_REF_CASES = sorted(f[:-4] for f in os.listdir(GOLD) if f.startswith("bf16ref_") and f.endswith(".npz"))
# the reference MODULE cast with .half() and run on CPU (tools/make_golden_r2.py, round 6): the reference script's own
# precision (/root/reference/CODON_X4/test.py:52,122-125)
FP16_REF_CASES = sorted(f[:-4] for f in os.listdir(GOLD) if f.startswith("fp16ref_") and f.endswith(".npz")
                        and f != "fp16ref_script_sizes.npz")

# the reference MODULE run forward AND backward in bfloat16 on CPU + its float64 twin (tools/make_golden_r4.py)
BF16_GRAD_CASES = sorted(f[:-4] for f in os.listdir(GOLD) if f.startswith("bf16grad_") and f.endswith(".npz"))


def bf16grad_inputs(v, B, H, W):
    """Inputs of variant v of a bf16grad_* fixture (tools/make_golden_r4.py::variant_inputs)."""
    if v == 0:
        return orc.kat_inputs(B, H, W)
    g = np.random.default_rng(1000 + v)
    x = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    y = torch.from_numpy(g.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32))
    return x, y


def load_case(name):
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    B, H, W = (int(v) for v in z["shape"])
    variant = str(z["variant"])
    wkind = str(z["weights"])
    if wkind == "kat":
        sd = orc.kat_state(variant)
    else:
        sd = orc.he_state(variant, seed={"he": 0, "he1": 1, "he2": 2}[wkind])
    x, y = orc.kat_inputs(B, H, W)
    return z, variant, sd, x, y


def target_for(x):
    """Same fixed target as tools/make_golden.py::target_for."""
    B, _, H, W = x.shape
    i = np.arange(H).reshape(1, 1, H, 1)
    j = np.arange(W).reshape(1, 1, 1, W)
    b = np.arange(B).reshape(B, 1, 1, 1)
    t = 0.5 + 0.45 * np.sin(0.37 * i + 0.11 * b) * np.cos(0.23 * j)
    return torch.from_numpy(t.astype(np.float32))


def rmse(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).pow(2).mean().sqrt())


def rel_rmse(a, b):
    b = torch.as_tensor(b).double()
    return rmse(a, b) / max(float(b.pow(2).mean().sqrt()), 1e-30)
