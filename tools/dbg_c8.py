"""Debug helper: pattern of bad elements, mask+acc dgrad."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice

dev = torch.device("cuda:0")
def rnd(shape, seed, s=1.0):
    return torch.from_numpy((np.random.default_rng(seed).standard_normal(size=shape) * s).astype(np.float32))
for dtype in (torch.bfloat16, torch.float16):
  for (B, H, W) in [(2, 21, 37), (1, 16, 64), (1, 8, 32)]:
    for (k, cin, cout) in [(3, 64, 64), (3, 128, 64), (5, 64, 64)]:
        w = rnd((cout, cin, k, k), 2, (2.0 / (k * k * cout)) ** 0.5).to(dtype).float().to(dev)
        gy = rnd((B, cout, H, W), 3).to(dtype).float().to(dev)
        act = rnd((B, cin, H, W), 4).to(dtype).float().to(dev)
        prev = rnd((B, cin, H, W), 5).to(dtype).float().to(dev)
        g0 = prev.clone()
        ops.conv2d(Slice(gy), ops.packed_weight(w, L.PACK_DGRAD), Slice(g0), k, relu_mask=Slice(act), accumulate=True)
        gyb, actb, wp = ops.from_nchw(gy, dtype), ops.from_nchw(act, dtype), ops.packed_weight(w, L.PACK_DGRAD, dtype)
        g1 = ops.from_nchw(prev, dtype)
        ops.conv2d(Slice(gyb), wp, Slice(g1), k, relu_mask=Slice(actb), accumulate=True)
        torch.cuda.synchronize()
        o = ops.to_nchw(g1).float()
        d = (o - g0).abs().nan_to_num(1e30)
        idx = (d > 0.1).nonzero()
        print(dtype, (B, H, W), (k, cin, cout), "bad:", len(idx), "b:", sorted(set(idx[:, 0].tolist())), "ch:", sorted(set(idx[:, 1].tolist())),
              "rows:", sorted(set(idx[:, 2].tolist())), "cols:", sorted(set(idx[:, 3].tolist())))
