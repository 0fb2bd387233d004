"""Run ON THE GPU BOX: does the one-image latency depend on how long the loop has been running (clock ramp)?  Times
consecutive blocks of 20 hipGraph replays.   b1_warm.py <fp16|fp32> <H> <W>"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from codon_amd import CODONNet
from codon_amd.graph import GraphedCODON
dt, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
m = CODONNet().cuda().eval()
if dt == "fp16":
    m = m.half()
x = torch.rand((1, 1, H, W), device="cuda"); y = torch.rand((1, 1, H, W), device="cuda")
if dt == "fp16":
    x, y = x.half(), y.half()
with torch.no_grad():
    gm = GraphedCODON(m, x, y)
    torch.cuda.synchronize()
    time.sleep(1.0)                      # let the chip idle first, as between two legs of bench.py
    out = []
    for blk in range(15):
        t0 = time.perf_counter()
        for _ in range(20):
            gm(x, y)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"{dt} 1x{H}x{W} graph replay, ms per forward in consecutive blocks of 20:", " ".join(f"{v:.3f}" for v in out))
