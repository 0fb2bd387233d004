"""One image per call, the reference script's own calling pattern (/root/reference/CODON_X4/test.py:116-125): the program
rocprofv3 runs directly (tools/probes/trace_b1.sh).   trace_b1.py <fp16|fp32|bf16> <H> <W> [iters]
Prints ms per forward (eager, and hipGraph replay when CODON_B1_GRAPH=1) -- under the profiler these are inflated; the
kernel table is what the trace is for."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from codon_amd import CODONNet

dt_name, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 30
torch.manual_seed(0)
m = CODONNet().cuda().eval()
if dt_name == "fp16":
    m = m.half()
elif dt_name == "bf16":
    m.set_compute_dtype(torch.bfloat16)
dtype = torch.float16 if dt_name == "fp16" else torch.float32
x = torch.rand((1, 1, H, W), device="cuda").to(dtype)
y = torch.rand((1, 1, H, W), device="cuda").to(dtype)
with torch.no_grad():
    for _ in range(5):
        m(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        m(x, y)
    torch.cuda.synchronize()
    print(f"{dt_name} 1x{H}x{W}: eager {(time.perf_counter() - t0) / iters * 1e3:.3f} ms/forward over {iters}")
    if os.environ.get("CODON_B1_GRAPH", "0") != "0":
        from codon_amd.graph import GraphedCODON
        gm = GraphedCODON(m, x, y)
        for _ in range(3):
            gm(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            gm(x, y)
        torch.cuda.synchronize()
        print(f"{dt_name} 1x{H}x{W}: hipGraph replay {(time.perf_counter() - t0) / iters * 1e3:.3f} ms/forward over {iters}")
