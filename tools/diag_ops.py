import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.nn.functional as F
from oracle import codon_oracle as orc
from tests.util import load_case, rel_rmse, target_for
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice
dev = torch.device("cuda:0")
z, variant, sd, x, y = load_case("kat0_x4_2x32x24")
tgt = target_for(x)
p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
taps = {}
out = orc.forward(p, x, y, taps)
for k, v in taps.items():
    if v.requires_grad: v.retain_grad()
gup = torch.sign(out.detach() - tgt) / out.numel()
out.backward(gup)
d = lambda t: t.detach().float().to(dev).contiguous()
for i in (4, 3):
    xc = (taps[f"blk{i-1}.out_c"]).detach().clone().requires_grad_(True)
    w4, w5, w6, wc = (sd[k + ".weight"] for k in ("conv4", "conv5", "conv6", "confuse_c"))
    R1 = F.relu(F.conv2d(xc, w4, None, 1, 2)); P1 = F.relu(F.conv2d(xc, w5, None, 1, 1))
    st = torch.cat((R1, P1), 1); st.retain_grad()
    R2 = F.relu(F.conv2d(st, w6, None, 1, 2)); R2.retain_grad()
    pre_c = F.conv2d(R2, wc)
    g_pre_c = taps[f"blk{i}.pre_c"].grad
    pre_c.backward(g_pre_c)
    print(f"blk{i}: fwd pre_c recompute vs tap {rel_rmse(pre_c, taps[f'blk{i}.pre_c']):.1e}")
    # HIP ops on torch inputs
    gpc = d(g_pre_c); r2d = d(R2); std = d(st)
    g_r2 = torch.empty((2, 128, 32, 24), device=dev); g_st = torch.empty((2, 128, 32, 24), device=dev)
    ops.conv2d(Slice(gpc), ops.packed_weight(d(wc), L.PACK_DGRAD), Slice(g_r2), 1, relu_mask=Slice(r2d))
    print(f"   confuse_c dgrad+mask {rel_rmse(g_r2.cpu(), R2.grad * (R2 > 0)):.2e}   (nonzero frac r2 {(R2>0).float().mean():.3f}, tiny positives {(R2.detach()[R2>0] < 1e-6).sum()})")
    g_r2_t = d(R2.grad * (R2 > 0))
    ops.conv2d(Slice(g_r2_t), ops.packed_weight(d(w6), L.PACK_DGRAD), Slice(g_st), 5, relu_mask=Slice(std))
    print(f"   conv6 dgrad+mask     {rel_rmse(g_st.cpu(), st.grad * (st > 0)):.2e}")
    g_st_t = d(st.grad * (st > 0))
    gx = torch.empty((2, 128, 32, 24), device=dev)
    ops.conv2d(Slice(g_st_t, 0, 64), ops.packed_weight(d(w4), L.PACK_DGRAD), Slice(gx, 64, 64), 5)
    ops.conv2d(Slice(g_st_t, 64, 64), ops.packed_weight(d(w5), L.PACK_DGRAD), Slice(gx, 64, 64), 3, accumulate=True)
    print(f"   conv4+conv5 dgrad    {rel_rmse(gx[:, 64:].cpu(), xc.grad):.2e}")
