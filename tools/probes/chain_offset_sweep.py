"""fp32 chained conv5x5-128 + 1x1 at 1 x 370 x 463: time against the distance between the input and the output buffer.
chain_offset_sweep.py <tree root>     (one arena; x at its 2 MiB-aligned start, the 128-channel output buffer D bytes later)"""
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
arena = torch.empty(700 * MiB, dtype=torch.uint8, device=dev)
base = (-arena.data_ptr()) % (2 * MiB)
nbytes = B * 128 * H * W * 4


def at(off):
    return arena[base + off: base + off + nbytes].view(torch.float32).view(B, 128, H, W)


x = at(0)
x.copy_(torch.relu(torch.randn((B, 128, H, W), device=dev)))
w5 = ops.packed_weight(torch.randn((128, 128, 5, 5), device=dev) * 0.02, L.PACK_FWD, torch.float32)
w1 = ops.packed_weight(torch.randn((64, 128, 1, 1), device=dev) * 0.1, L.PACK_CHAIN1X1, torch.float32)


def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print(os.path.basename(os.path.abspath(sys.argv[1])))
for d in [84, 86, 88, 90, 92, 96, 100, 104, 128, 130, 132, 136, 216, 218, 220, 222, 224, 228, 256, 260, 512, 516]:
    o = at(d * MiB)
    print(f"  out - x = {d:4d} MiB: hi half {t(lambda: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64, 64))):.3f}  "
          f"lo half {t(lambda: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 0, 64))):.3f} ms", flush=True)
for k in range(0, 17):
    d = 216 * MiB + k * 256 * 1024
    o = at(d)
    print(f"  out - x = 216 MiB + {k * 256:5d} KiB: {t(lambda: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64, 64))):.3f} ms", flush=True)
