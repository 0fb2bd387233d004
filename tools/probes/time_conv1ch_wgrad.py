"""Run ON THE GPU BOX: time of codon_conv1ch_wgrad on 16-bit tensors at the C2 shape, and a checksum of dw (bit-identity
between build variants)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from codon_amd import ops
from codon_amd.ops import Slice
B, H, W = 32, 480, 640
dev = torch.device("cuda:0")
torch.manual_seed(0)
a = ops.from_nchw(torch.relu(torch.randn((B, 64, H, W), device=dev)), torch.bfloat16)
s = torch.rand((B, 1, H, W), device=dev)
dw = torch.empty((64, 1, 3, 3), device=dev)
for _ in range(3):
    ops.conv1ch_wgrad(Slice(a), s, dw, flip=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.conv1ch_wgrad(Slice(a), s, dw, flip=False)
e1.record(); torch.cuda.synchronize()
print(f"conv1ch_wgrad bf16: {e0.elapsed_time(e1) / 20:.3f} ms   sum {float(dw.double().sum()):.10e}  abs {float(dw.double().abs().sum()):.10e}")
