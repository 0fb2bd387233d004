"""16-bit one-image convs against the image height (W = 463: 15 tile columns): where the rounds of workgroups fall.
c8_rounds_sweep.py [fp16|bf16]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice

dev = torch.device("cuda:0")
dt = torch.float16 if (len(sys.argv) < 2 or sys.argv[1] == "fp16") else torch.bfloat16
torch.manual_seed(0)
w5 = ops.packed_weight(torch.randn((128, 128, 5, 5), device=dev) * 0.02, L.PACK_FWD, dt)
w1 = ops.packed_weight(torch.randn((64, 128, 1, 1), device=dev) * 0.1, L.PACK_CHAIN1X1, dt)
w564 = ops.packed_weight(torch.randn((64, 64, 5, 5), device=dev) * 0.02, L.PACK_FWD, dt)
w364 = ops.packed_weight(torch.randn((64, 64, 3, 3), device=dev) * 0.02, L.PACK_FWD, dt)


def t(fn, n=30):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


W = 463
print(dt)
for H in (120, 136, 200, 208, 240, 272, 304, 336, 370, 400, 440, 480, 546, 600, 680, 760, 820):
    x = ops.from_nchw(torch.relu(torch.randn((1, 128, H, W), device=dev)), dt)
    o = ops.new_act(1, 128, H, W, dt, dev)
    n8 = 15 * ((H + 7) // 8)

    def pair(f):
        def g():
            with ops.conv_pair(dev):
                f(0); f(1)
        return g

    chain = lambda k=0: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64 * k, 64))
    c5 = lambda k=0: ops.conv2d(Slice(x, 64 * k, 64), w564, Slice(o, 64 * k, 64), 5, relu=True)
    c3 = lambda k=0: ops.conv2d(Slice(x, 64 * k, 64), w364, Slice(o, 64 * k, 64), 3, relu=True)
    print(f"  H {H:4d}  tiles(8x32) {n8:5d}  chain {t(chain):6.1f}  pair {t(pair(chain)):6.1f}   conv5x5-64 {t(c5):5.1f}  pair {t(pair(c5)):5.1f}"
          f"   conv3x3-64 {t(c3):5.1f}  pair {t(pair(c3)):5.1f} us", flush=True)
