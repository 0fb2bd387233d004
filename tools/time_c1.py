"""BASELINE configs[0] on the GPU (1 x 128 x 128, fp32 exact): 20 forwards, for a rocprofv3 kernel trace of where the
4.6 ms go (run as: rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 tools/time_c1.py)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codon_amd import CODONNet
torch.manual_seed(0)
m = CODONNet().cuda().eval()
x, y = torch.rand(1, 1, 128, 128, device="cuda"), torch.rand(1, 1, 128, 128, device="cuda")
with torch.no_grad():
    for _ in range(3):
        m(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        m(x, y)
    torch.cuda.synchronize()
print(f"1x128x128 fp32: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/forward")
from codon_amd.graph import GraphedCODON
g = GraphedCODON(m, x, y)
for _ in range(3):
    g(x, y)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50):
    g(x, y)
torch.cuda.synchronize()
print(f"1x128x128 fp32, hipGraph replay: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms/forward")
