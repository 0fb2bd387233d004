"""What the host-side copies add to a forward (the nn.Module boundary takes DEVICE tensors -- the reference script moves every
image itself, test.py:116-125 -- so this is never bench.py's `value`):  numpy -> .cuda() -> model -> .cpu().numpy()."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from codon_amd import CODONNet

torch.manual_seed(0)
dev = torch.device("cuda:0")


def run(m, B, H, W, half, n, pinned):
    g = np.random.default_rng(0)
    xs = g.random((B, 1, H, W), dtype=np.float32)
    ys = g.random((B, 1, H, W), dtype=np.float32)
    xp = torch.from_numpy(xs).pin_memory() if pinned else torch.from_numpy(xs)
    yp = torch.from_numpy(ys).pin_memory() if pinned else torch.from_numpy(ys)

    def dev_only():
        return m(xd, yd)

    def incl():
        xd_ = xp.to(dev, non_blocking=pinned)
        yd_ = yp.to(dev, non_blocking=pinned)
        if half:
            xd_, yd_ = xd_.half(), yd_.half()
        return m(xd_, yd_).float().cpu().numpy()

    xd, yd = torch.from_numpy(xs).to(dev), torch.from_numpy(ys).to(dev)
    if half:
        xd, yd = xd.half(), yd.half()
    out = []
    with torch.no_grad():
        for f in (dev_only, incl):
            for _ in range(max(3, n // 4)):
                f()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                f()
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / n * 1e3)
    return out


m32 = CODONNet().to(dev).eval()
m16 = CODONNet().to(dev).eval().half()
for (B, H, W, half, n, pinned) in ((1, 370, 463, True, 50, False), (1, 370, 463, True, 50, True), (1, 128, 128, False, 50, False),
                                   (32, 480, 640, False, 3, True)):
    a, b = run(m16 if half else m32, B, H, W, half, n, pinned)
    print(f"{B} x {H} x {W} {'fp16' if half else 'fp32'} ({'pinned' if pinned else 'pageable'} host buffers): device tensors in and out "
          f"{a:.3f} ms, numpy in and out {b:.3f} ms (+{b - a:.3f})", flush=True)
