"""Does the physical placement of the activation buffers decide the speed of the fp32 one-image forward at 370 x 463?
perturb_b1.py <tree root> <KiB to hipMalloc (and keep) before anything else>   ->  ms / forward
The conv5x5-128 kernels read 128 channel planes 685 KB apart per tile: a TLB-heavy pattern, so the page-table fragment size the
driver could give each buffer matters.  A raw hipMalloc ahead of torch's allocations shifts every later physical placement."""
import ctypes
import os
import sys
import time

root, kib = sys.argv[1], int(sys.argv[2])
sys.path.insert(0, root)
import torch

torch.cuda.init()
hip = ctypes.CDLL("libamdhip64.so")
keep = ctypes.c_void_p()
if kib:
    assert hip.hipMalloc(ctypes.byref(keep), ctypes.c_size_t(kib * 1024)) == 0
from codon_amd import CODONNet

torch.manual_seed(0)
m = CODONNet().cuda().eval()
x = torch.rand((1, 1, 370, 463), device="cuda")
y = torch.rand((1, 1, 370, 463), device="cuda")
with torch.no_grad():
    for _ in range(5):
        m(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        m(x, y)
    torch.cuda.synchronize()
print(f"{os.path.basename(os.path.abspath(root))} perturb {kib} KiB: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms/forward", flush=True)
