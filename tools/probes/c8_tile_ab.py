"""Round 6 (needs tools/probes/c8_4x32_tiles_strip_statistics_experiment.patch applied): the 16-bit chained conv on 8 x 32 against 4 x 32 tiles (CODON_C8_CHAIN_TILE), alone and as a pair, over image
heights at W = 463, and the one-image forward with the rule on / forced to 8.  c8_tile_ab.py [fp16|bf16]"""
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


def t(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


W = 463
print(dt)
for H in (128, 200, 247, 304, 370, 375, 440, 480, 600, 760):
    x = ops.from_nchw(torch.relu(torch.randn((1, 128, H, W), device=dev)), dt)
    o = ops.new_act(1, 128, H, W, dt, dev)
    nt = ops.cac_fused_tiles(H, W)
    pool = [torch.empty((1, 2, H, W), device=dev) for _ in range(2)]
    part = torch.empty((1, nt, 128, 2), device=dev)
    chain = lambda k=0: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64 * k, 64), stats=(pool[k], part, 64 * k))

    def pair():
        with ops.conv_pair(dev):
            chain(0); chain(1)

    r = {}
    for tile in ("8", "4"):
        os.environ["CODON_C8_CHAIN_TILE"] = tile
        r[tile] = (t(chain), t(pair))
    os.environ.pop("CODON_C8_CHAIN_TILE")
    auto = (t(chain), t(pair))
    n8, n4 = 15 * ((H + 7) // 8), 15 * ((H + 3) // 4)
    print(f"  H {H:4d}  n8 {n8:5d} n4 {n4:5d}   lone 8x32 {r['8'][0]:6.1f}  4x32 {r['4'][0]:6.1f}  rule {auto[0]:6.1f}    pair 8x32 {r['8'][1]:6.1f}  "
          f"4x32 {r['4'][1]:6.1f}  rule {auto[1]:6.1f} us", flush=True)
