"""Run ON THE GPU BOX: where the HOST time of an eager one-image forward goes (cProfile over 300 forwards at a size whose GPU
time is far below the host time).  usage: host_profile.py [fp16|fp32] [H W]"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from codon_amd import CODONNet
dt = sys.argv[1] if len(sys.argv) > 1 else "fp16"
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (128, 128)
m = CODONNet().cuda().eval()
if dt == "fp16":
    m = m.half()
x = torch.rand((1, 1, H, W), device="cuda"); y = torch.rand((1, 1, H, W), device="cuda")
if dt == "fp16":
    x, y = x.half(), y.half()
with torch.no_grad():
    for _ in range(10): m(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300): m(x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"{dt} 1x{H}x{W}: host issue time {(t1 - t0) / 300 * 1e3:.3f} ms/forward, with sync {(time.perf_counter() - t0) / 300 * 1e3:.3f}")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300): m(x, y)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
