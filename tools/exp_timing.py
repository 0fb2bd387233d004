"""Per-workgroup phase timestamps of the fp32 conv kernel (debug build: make -C codon_amd/csrc EXTRA=-DCODON_TIMING).
Prints the prologue / main loop / epilogue durations per workgroup."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
k, ci, co = (int(v) for v in sys.argv[1:4])
dt = torch.bfloat16 if len(sys.argv) > 4 and sys.argv[4] == 'bf16' else torch.float32
dt = torch.bfloat16 if len(sys.argv) > 4 and sys.argv[4] == 'bf16' else torch.float32
B, H, W = int(os.environ.get("B", 32)), 480, 640
dev = torch.device("cuda:0")
th = 8
nblk = ((W + 31) // 32) * ((H + th - 1) // th) * B
dbg = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
os.environ["CODON_DBG_PTR"] = hex(dbg.data_ptr())
from codon_amd import ops
from codon_amd.ops import Slice
x = torch.randn((B, ci, H, W), device=dev).to(dt)
w = torch.randn((co, ci, k, k), device=dev) * 0.05
wp = ops.packed_weight(w, dtype=dt)
y = torch.empty((B, co, H, W), device=dev, dtype=dt)
for _ in range(2):
    ops.conv2d(Slice(x), wp, Slice(y), k, relu=True)
torch.cuda.synchronize()
d = dbg.cpu().numpy()
t = d[:, :5].astype(np.float64) * 10.0 / 1e3     # us (100 MHz clock)
t0 = t[:, 0].min()
print(f"kernel span {t[:, 4].max() - t0:.1f} us, {nblk} workgroups")
pro, bar, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
for nm, v in (("prologue (start -> staged)", pro), ("first barrier", bar), ("main loop", loop), ("epilogue", epi),
              ("whole workgroup", t[:, 4] - t[:, 0])):
    print(f"{nm:28s} mean {v.mean():8.2f} us   p10 {np.percentile(v, 10):8.2f}  p50 {np.percentile(v, 50):8.2f}  p90 {np.percentile(v, 90):8.2f}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    ops.conv2d(Slice(x), wp, Slice(y), k, relu=True)
e1.record(); torch.cuda.synchronize()
print(f"avg launch {e0.elapsed_time(e1) / 5:.3f} ms")
