"""fp32 one-image convs against the image height (W = 463: 15 tile columns), in the tile mode the environment selects
(tools/probes/grid_mode_build.sh).  grid_mode_sweep.py <label>"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice

dev = torch.device("cuda:0")
torch.manual_seed(0)
w5 = ops.packed_weight(torch.randn((128, 128, 5, 5), device=dev) * 0.02, L.PACK_FWD, torch.float32)
w1 = ops.packed_weight(torch.randn((64, 128, 1, 1), device=dev) * 0.1, L.PACK_CHAIN1X1, torch.float32)
w564 = ops.packed_weight(torch.randn((64, 64, 5, 5), device=dev) * 0.02, L.PACK_FWD, torch.float32)
w364 = ops.packed_weight(torch.randn((64, 64, 3, 3), device=dev) * 0.02, L.PACK_FWD, torch.float32)


def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


W = 463
print(sys.argv[1])
for H in (200, 208, 240, 272, 304, 336, 370, 400, 440, 480, 546, 600, 680, 760, 820):
    x = torch.relu(torch.randn((1, 128, H, W), device=dev))
    o = torch.empty((1, 128, H, W), device=dev)
    n8 = 15 * ((H + 7) // 8)
    a = t(lambda: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64, 64)))
    b = t(lambda: ops.conv2d(Slice(x, 0, 64), w564, Slice(o, 0, 64), 5, relu=True))
    c = t(lambda: ops.conv2d(Slice(x, 0, 64), w364, Slice(o, 64, 64), 3, relu=True))
    print(f"  H {H:4d}  tiles(8x32) {n8:5d}  chain {a:.3f}  conv5x5-64 {b:.3f}  conv3x3-64 {c:.3f} ms", flush=True)
