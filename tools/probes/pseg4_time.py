"""Chained fp32 conv5x5-128 + 1x1 at 32 x 480 x 640: 8 x 32 tiles two per CU (product) vs 16 x 32 tiles one per CU
(tools/probes/pseg4_build.sh; CODON_AMD_LIB must point at that library)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice

dev = torch.device("cuda:0")
B, H, W = 32, 480, 640
torch.manual_seed(0)
x = torch.relu(torch.randn((B, 128, H, W), device=dev))
o = torch.empty((B, 64, H, W), device=dev)
r = torch.randn((B, 64, H, W), device=dev)
w5 = ops.packed_weight(torch.randn((128, 128, 5, 5), device=dev) * 0.02, L.PACK_FWD, torch.float32)
w1 = ops.packed_weight(torch.randn((64, 128, 1, 1), device=dev) * 0.1, L.PACK_CHAIN1X1, torch.float32)


def t(n=6):
    f = lambda: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o), residual=Slice(r))
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


os.environ.pop("CODON_PROBE_PSEG4", None)
a = t(); ya = o.clone()
os.environ["CODON_PROBE_PSEG4"] = "1"
b = t(); yb = o.clone()
os.environ.pop("CODON_PROBE_PSEG4", None)
a2 = t()
os.environ["CODON_PROBE_PSEG4"] = "1"
b2 = t()
print(f"8 x 32 two per CU: {a:.2f} / {a2:.2f} ms   16 x 32 one per CU: {b:.2f} / {b2:.2f} ms   same bits: {torch.equal(ya, yb)}")
