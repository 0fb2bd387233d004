"""fp32 chained conv5x5-128 + 1x1 at 1 x 370 x 463: time against the address of the packed 1x1 weights modulo 4 KiB.
chain_w1_sweep.py <tree root>"""
import os
import sys

sys.path.insert(0, sys.argv[1])
import torch
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice

dev = torch.device("cuda:0")
B, H, W = 1, 370, 463
MiB = 1 << 20
torch.manual_seed(0)
x = torch.relu(torch.randn((B, 128, H, W), device=dev))
o = torch.empty((B, 128, H, W), device=dev)
o64 = torch.empty((B, 64, H, W), device=dev)
r64 = torch.randn((B, 64, H, W), device=dev)
w5 = ops.packed_weight(torch.randn((128, 128, 5, 5), device=dev) * 0.02, L.PACK_FWD, torch.float32)
w1 = ops.packed_weight(torch.randn((64, 128, 1, 1), device=dev) * 0.1, L.PACK_CHAIN1X1, torch.float32)
arena = torch.empty(8 * MiB, dtype=torch.uint8, device=dev)
base = (-arena.data_ptr()) % (2 * MiB)


def place(t, off):
    v = arena[base + off: base + off + t.numel() * 4].view(torch.float32).view(t.shape)
    v.copy_(t)
    return v


def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(os.path.basename(os.path.abspath(sys.argv[1])))
for off in list(range(0, 0x1000, 0x200)) + [0x80, 0x100, 0x180, 0x680, 0x10600, 0x100600]:
    b = place(w1, off)
    print(f"  w1 at 2 MiB-aligned + {off:#8x}: block form {t(lambda: ops.conv_chain1x1(Slice(x), w5, b, Slice(o, 64, 64))):.3f}"
          f"   trunk form (64/64 + residual) {t(lambda: ops.conv_chain1x1(Slice(x), w5, b, Slice(o64), residual=Slice(r64))):.3f} ms", flush=True)
