"""The two one-image-per-call numbers of the bench line (config0_on_gpu, script_pattern_on_gpu fp16 370 x 463), same protocol
(steady state: mean of 50 calls after 20 warm-up calls), eager and hipGraph replay -- a 20-second run instead of the bench."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from codon_amd import CODONNet
from codon_amd.graph import GraphedCODON

dev = torch.device("cuda:0")


def lat(fn, n=50, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e3


torch.manual_seed(0)
cases = [("fp32", 128, 128), ("fp16", 370, 463), ("fp16", 247, 343), ("fp32", 370, 463)]
if len(sys.argv) > 1:
    cases = [c for c in cases if f"{c[0]}_{c[1]}x{c[2]}" in sys.argv[1:]]
for dt_name, H, W in cases:
    m = CODONNet().to(dev)
    m = (m.half() if dt_name == "fp16" else m).eval()
    dtype = torch.float16 if dt_name == "fp16" else torch.float32
    x, y = torch.rand((1, 1, H, W), device=dev).to(dtype), torch.rand((1, 1, H, W), device=dev).to(dtype)
    with torch.no_grad():
        gm = GraphedCODON(m, x, y)
        e = [lat(lambda: m(x, y)) for _ in range(2)]
        g = [lat(lambda: gm(x, y)) for _ in range(2)]
    print(f"{dt_name} 1x{H}x{W}: eager {e[0]:.3f} / {e[1]:.3f} ms   hipGraph {g[0]:.3f} / {g[1]:.3f} ms", flush=True)
