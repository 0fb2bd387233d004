"""16-bit chained conv + statistics epilogue, one launch: 32 x 480 x 640 (the batch-32 forward's dominant launch) and one
370 x 463 image as a pair.  Run it under two libraries (CODON_AMD_LIB) on the same box for an A/B.  chain_stats_ab.py [bf16|fp16]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice

dev = torch.device("cuda:0")
dt = torch.float16 if (len(sys.argv) > 1 and sys.argv[1] == "fp16") else torch.bfloat16
torch.manual_seed(0)
w5 = ops.packed_weight(torch.randn((128, 128, 5, 5), device=dev) * 0.02, L.PACK_FWD, dt)
w1 = ops.packed_weight(torch.randn((64, 128, 1, 1), device=dev) * 0.1, L.PACK_CHAIN1X1, dt)


def t(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (B, H, W, n) in ((32, 480, 640, 10), (1, 370, 463, 50)):
    x = ops.from_nchw(torch.relu(torch.randn((B, 128, H, W), device=dev)), dt)
    o = ops.new_act(B, 128, H, W, dt, dev)
    nt = ops.cac_fused_tiles(H, W)
    pool = [torch.empty((B, 2, H, W), device=dev) for _ in range(2)]
    part = torch.empty((B, nt, 128, 2), device=dev)
    chain = lambda k=0: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64 * k, 64), stats=(pool[k], part, 64 * k))
    plain = lambda k=0: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64 * k, 64))

    def pair():
        with ops.conv_pair(dev):
            chain(0); chain(1)

    r = [t(chain, n), t(plain, n), t(chain, n), t(plain, n)]
    extra = f"   pair with stats {t(pair, n):8.1f} / {t(pair, n):8.1f}" if B == 1 else ""
    print(f"{dt} {B}x{H}x{W}: with stats {r[0]:9.1f} / {r[2]:9.1f} us   without {r[1]:9.1f} / {r[3]:9.1f} us{extra}", flush=True)
