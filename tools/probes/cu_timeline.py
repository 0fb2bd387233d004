"""Per-CU timelines of the fp32 conv kernel from a -DCODON_TIMING build (tools/ab_build.sh timing conv_mfma_f32.hip
-DCODON_TIMING): for every CU, how many of its resident workgroups are inside the main loop at a time.  If the co-resident
workgroups run in phase, the CU spends a visible share of the launch with NONE of them feeding the matrix pipe.
usage: cu_timeline.py k cin cout"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
k, ci, co = (int(v) for v in sys.argv[1:4])
B, H, W = int(os.environ.get("B", 32)), 480, 640
dev = torch.device("cuda:0")
nblk = ((W + 31) // 32) * ((H + 7) // 8) * B
dbg = torch.zeros((nblk, 8), dtype=torch.int64, device=dev)
os.environ["CODON_DBG_PTR"] = hex(dbg.data_ptr())
from codon_amd import ops
from codon_amd.ops import Slice
x = torch.randn((B, ci, H, W), device=dev)
w = torch.randn((co, ci, k, k), device=dev) * 0.05
wp = ops.packed_weight(w, dtype=torch.float32)
y = torch.empty((B, co, H, W), device=dev)
for _ in range(2):
    ops.conv2d(Slice(x), wp, Slice(y), k, relu=True)
torch.cuda.synchronize()
d = dbg.cpu().numpy()
t = d[:, :5].astype(np.float64) * 0.01          # us (100 MHz clock)
t -= t[:, 0].min()
hw, xcc = d[:, 6], d[:, 7] & 0xF
cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)).astype(np.int64)
ids = np.unique(cu)
span = t[:, 4].max()
print(f"{nblk} workgroups on {len(ids)} CUs, kernel span {span:.1f} us; phases per workgroup (us): "
      f"prologue {np.mean(t[:,2]-t[:,0]):.2f}  main {np.mean(t[:,3]-t[:,2]):.2f}  epilogue {np.mean(t[:,4]-t[:,3]):.2f}")
hist = np.zeros(9)
res_hist = np.zeros(9)
for c in ids:
    m = cu == c
    ev = [(a, 1, 0) for a in t[m, 2]] + [(b, -1, 0) for b in t[m, 3]] + [(a, 0, 1) for a in t[m, 0]] + [(b, 0, -1) for b in t[m, 4]]
    ev.sort()
    n = r = 0
    last = 0.0
    for tm, dn, dr in ev:
        hist[min(n, 8)] += tm - last
        res_hist[min(r, 8)] += tm - last
        last = tm
        n += dn
        r += dr
    hist[0] += span - last
    res_hist[0] += span - last
tot = hist.sum()
print("share of CU time with n workgroups in the main loop:", " ".join(f"{i}:{hist[i] / tot:.3f}" for i in range(6)))
print("share of CU time with n workgroups resident:        ", " ".join(f"{i}:{res_hist[i] / tot:.3f}" for i in range(6)))
# phase alignment: spread of main-loop entry times among the workgroups of a CU, modulo the mean workgroup duration
dur = np.mean(t[:, 4] - t[:, 0])
ph = []
for c in ids[:64]:
    v = np.sort(t[cu == c, 2] % dur)
    ph.append(np.std(v) / dur)
print(f"mean workgroup duration {dur:.1f} us; std of (main-loop entry mod duration) / duration over the first 64 CUs: {np.mean(ph):.3f} (uniform = 0.289)")
