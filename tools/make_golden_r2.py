#!/usr/bin/env python3
"""Round-2 additions to tests/golden/ (same rules as tools/make_golden.py: imports the REFERENCE's own Python on CPU,
runs only in the build container, writes data only).  The round-1 fixtures are left untouched.

  he2_x16_1x21x27   He-init x16 forward + autograd gradients (strict per-tensor gradient case; KAT-0's
                     low-discrepancy weights put pre-activations at 1e-9 where a ReLU mask may legitimately flip)
  he1_x4_2x18x22    He-init x4 forward + gradients, batch 2, ragged size
  bf16ref_*         the reference MODULE cast to bfloat16 and run on CPU (net.bfloat16()(x.bfloat16(), y.bfloat16())):
                     pins the bf16 tolerance of the HIP bf16 path to the reference's own bf16 behaviour, beside
                     the fp64 output of the same net
  fp16ref_*         (round 6) the same for float16 -- the ONLY precision the reference script runs
                     (/root/reference/CODON_X4/test.py:52,122-125: model.cuda().half(), inputs .half()): the module
                     cast with .half() and run on CPU, beside its fp32 and fp64 outputs; the three bf16ref shapes plus
                     one image of the script's own size (Middlebury "Art", 370 x 463)
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

GRAD_CASES = [
    ("he2_x16_1x21x27", "x16", "he2", (1, 21, 27)),
    ("he1_x4_2x18x22", "x4", "he1", (2, 18, 22)),
]
BF16_CASES = [
    ("bf16ref_he0_x4_2x24x20", "x4", "he", (2, 24, 20)),
    ("bf16ref_he1_x4_1x40x56", "x4", "he1", (1, 40, 56)),
    ("bf16ref_he0_x16_1x33x9", "x16", "he", (1, 33, 9)),
]
FP16_CASES = [
    ("fp16ref_he0_x4_2x24x20", "x4", "he", (2, 24, 20)),
    ("fp16ref_he1_x4_1x40x56", "x4", "he1", (1, 40, 56)),
    ("fp16ref_he0_x16_1x33x9", "x16", "he", (1, 33, 9)),
    ("fp16ref_he2_x4_1x370x463", "x4", "he2", (1, 370, 463)),
    ("fp16ref_kat0_x4_2x32x24", "x4", "kat", (2, 32, 24)),
    ("fp16ref_kat0_x4_1x1x1", "x4", "kat", (1, 1, 1)),
]
# the Middlebury image sizes the reference script feeds (one image per call, test.py:116-125) with the random inputs of
# tests/test_gpu_forward.py::test_forward_at_the_reference_scripts_image_sizes: every SUB-th pixel of the reference module's
# fp16 / fp32 / fp64 outputs (a full image is 1.5 MB; the GPU box's CPU runs fp16 convs two orders of magnitude slower than
# this container's, so the yardstick is recorded here instead of being recomputed there)
SCRIPT_SIZES = [(370, 463), (375, 450), (247, 343)]
SCRIPT_SUB = 7


def script_inputs(H, W):
    g = np.random.default_rng(H)
    x = torch.from_numpy(g.random((1, 1, H, W), dtype=np.float32))
    y = torch.from_numpy((g.integers(0, 256, size=(1, 1, H, W)) / 255.0).astype(np.float32))
    return x, y


def main():
    torch.set_num_threads(8)
    nets = {}

    def net_for(variant):
        if variant not in nets:
            nets[variant] = mg.load_reference(variant)
        return nets[variant]

    for name, variant, wkind, (B, H, W) in GRAD_CASES:
        net = net_for(variant)
        sd = orc.he_state(variant, seed=SEEDS[wkind])
        x, y = orc.kat_inputs(B, H, W)
        out, _ = mg.run_reference(net, sd, x, y, False)
        rec = {"out": out.numpy(), "shape": np.array([B, H, W]), "variant": variant, "weights": wkind}
        tgt = mg.target_for(x)
        loss, gs = mg.ref_grads(net, sd, x, y, tgt)
        rec["loss"] = np.float64(loss)
        for k, g in gs.items():
            stride, s = mg.sub(g)
            rec["grad." + k] = s
            rec["gradstride." + k] = np.int64(stride)
            rec["gradnorm." + k] = np.float64(g.double().norm())
            rec["gradsum." + k] = np.float64(g.double().sum())
        for p in net.parameters():
            p.requires_grad_(False)
        with torch.no_grad():
            rec["out_fp64"] = net.double()(x.double(), y.double()).numpy()
        net.float()
        np.savez_compressed(os.path.join(mg.GOLD, name + ".npz"), **rec)
        print(name, "loss", loss, "out std", float(out.std()))

    for name, variant, wkind, (B, H, W) in BF16_CASES:
        net = net_for(variant)
        sd = orc.he_state(variant, seed=SEEDS[wkind])
        x, y = orc.kat_inputs(B, H, W)
        net.load_state_dict(sd, strict=True)
        net.eval()
        with torch.no_grad():
            o32 = net(x, y)
            o64 = net.double()(x.double(), y.double())
            ob = net.bfloat16()(x.bfloat16(), y.bfloat16())
        net.float()
        net.load_state_dict(sd, strict=True)        # undo the bf16 rounding of the parameters
        rec = {"shape": np.array([B, H, W]), "variant": variant, "weights": wkind, "out": o32.numpy(),
               "out_fp64": o64.numpy(), "out_bf16": ob.float().numpy()}
        np.savez_compressed(os.path.join(mg.GOLD, name + ".npz"), **rec)
        e = float((ob.double() - o64).pow(2).mean().sqrt() / o64.pow(2).mean().sqrt())
        print(f"{name}: reference bf16-vs-fp64 rel-RMSE {e:.3e}")

    for name, variant, wkind, (B, H, W) in FP16_CASES:
        net = net_for(variant)
        sd = orc.kat_state(variant) if wkind == "kat" else orc.he_state(variant, seed=SEEDS[wkind])
        x, y = orc.kat_inputs(B, H, W)
        net.float()
        net.load_state_dict(sd, strict=True)
        net.eval()
        with torch.no_grad():
            o32 = net(x, y)
            o64 = net.double()(x.double(), y.double())
            oh = net.half()(x.half(), y.half())          # test.py:52 + :122-123 on the CPU
        net.float()
        net.load_state_dict(sd, strict=True)        # undo the fp16 rounding of the parameters
        assert oh.dtype == torch.float16
        rec = {"shape": np.array([B, H, W]), "variant": variant, "weights": wkind, "out": o32.numpy(),
               "out_fp64": o64.numpy(), "out_fp16": oh.numpy()}
        np.savez_compressed(os.path.join(mg.GOLD, name + ".npz"), **rec)
        e = float((oh.double() - o64).pow(2).mean().sqrt() / o64.pow(2).mean().sqrt())
        print(f"{name}: reference fp16-vs-fp64 rel-RMSE {e:.3e}")

    rec = {"sub": np.int64(SCRIPT_SUB), "sizes": np.array(SCRIPT_SIZES)}
    net = net_for("x4")
    for H, W in SCRIPT_SIZES:
        sd = orc.he_state("x4", seed=70 + H)
        x, y = script_inputs(H, W)
        net.float()
        net.load_state_dict(sd, strict=True)
        net.eval()
        with torch.no_grad():
            o32 = net(x, y)
            o64 = net.double()(x.double(), y.double())
            oh = net.half()(x.half(), y.half())
        net.float()
        net.load_state_dict(sd, strict=True)
        tag = f"{H}x{W}"
        rec[tag + ".out_fp64_sub"] = o64.numpy().reshape(-1)[::SCRIPT_SUB]
        rec[tag + ".out_fp32_sub"] = o32.numpy().reshape(-1)[::SCRIPT_SUB]
        rec[tag + ".out_fp16_sub"] = oh.numpy().reshape(-1)[::SCRIPT_SUB]
        e = float((oh.double() - o64).pow(2).mean().sqrt() / o64.pow(2).mean().sqrt())
        rec[tag + ".ref_err_full"] = np.float64(e)
        rec[tag + ".x00"] = np.float32(x[0, 0, 0, 0])          # the inputs are regenerated from the seed: a guard against a
        rec[tag + ".y_last"] = np.float32(y[0, 0, -1, -1])     # numpy whose Generator streams differ
        print(f"fp16ref_script_sizes {tag}: reference fp16-vs-fp64 rel-RMSE {e:.3e}")
    np.savez_compressed(os.path.join(mg.GOLD, "fp16ref_script_sizes.npz"), **rec)


if __name__ == "__main__":
    main()
