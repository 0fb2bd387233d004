"""fp32 chained conv5x5-128 + 1x1 at 1 x 370 x 463: time against the ADDRESS of the packed 5x5 weights (and of the 1x1 weights).
chain_weight_sweep.py <tree root>"""
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
w5 = ops.packed_weight(torch.randn((128, 128, 5, 5), device=dev) * 0.02, L.PACK_FWD, torch.float32)
w1 = ops.packed_weight(torch.randn((64, 128, 1, 1), device=dev) * 0.1, L.PACK_CHAIN1X1, torch.float32)
arena = torch.empty(80 * MiB, dtype=torch.uint8, device=dev)
base = (-arena.data_ptr()) % (32 * MiB)


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


print(os.path.basename(os.path.abspath(sys.argv[1])), f"w5 {w5.data_ptr():#x} ({w5.numel() * 4} B)  w1 {w1.data_ptr():#x}  x {x.data_ptr():#x}")
print(f"  torch's own placement: {t(lambda: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64, 64))):.3f} ms")
for k in range(0, 16):
    for sub in (0, 2 * MiB - 0x1600):
        a = place(w5, k * 2 * MiB + sub)
        print(f"  w5 at 32 MiB-aligned + {k * 2:2d} MiB + {sub:#9x}: {t(lambda: ops.conv_chain1x1(Slice(x), a, w1, Slice(o, 64, 64))):.3f} ms", flush=True)
a = place(w5, 0)
for off in (40 * MiB, 40 * MiB + 0x3600, 40 * MiB + 0x53600, 42 * MiB - 0x1000, 44 * MiB + 512):
    b = place(w1, off)
    print(f"  w1 at + {off:#x}: {t(lambda: ops.conv_chain1x1(Slice(x), a, b, Slice(o, 64, 64))):.3f} ms", flush=True)
