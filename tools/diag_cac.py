import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import codon_oracle as orc
from tests.util import load_case, rel_rmse, target_for
from codon_amd import ops
from codon_amd.ops import Slice
dev = torch.device("cuda:0")
z, variant, sd, x, y = load_case("kat0_x4_2x32x24")
tgt = target_for(x)
for dt in (torch.float32,):
    p = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
    taps = {}
    out = orc.forward(p, x.to(dt), y.to(dt), taps)
    for k, v in taps.items():
        if v.requires_grad: v.retain_grad()
    gup = torch.sign(out.detach() - tgt) / out.numel()
    out.backward(gup.to(dt))
p64 = {k: v.double().clone().requires_grad_(True) for k, v in sd.items()}
taps64 = {}
out64 = orc.forward(p64, x.double(), y.double(), taps64)
for k, v in taps64.items():
    if v.requires_grad: v.retain_grad()
out64.backward(gup.double())
B, _, H, W = x.shape
for i in range(5):
    pre, pre_c = taps[f"blk{i}.pre"], taps[f"blk{i}.pre_c"]
    g_out, g_outc = taps[f"blk{i}.out"].grad, taps[f"blk{i}.out_c"].grad
    d = lambda t: t.detach().float().to(dev).contiguous()
    p2 = torch.cat((d(pre), d(pre_c)), 1).contiguous()
    goc = torch.cat((d(g_out), d(g_outc)), 1).contiguous()
    nt = ops.cac_stats_tiles(H, W)
    pooled = torch.empty((B, 2, H, W), device=dev); partials = torch.empty((B, nt, 128, 2), device=dev)
    chd = torch.empty((B, 64), device=dev); pools = torch.empty((B, 2, 128), device=dev); spd = torch.empty((B, 1, H, W), device=dev)
    w1, b1, w2, b2 = (d(sd[f"attention_c{i}.mlp.{j}.{n}"]) for j, n in ((1, "weight"), (1, "bias"), (3, "weight"), (3, "bias")))
    ws = d(sd[f"attention_s{i}.spatial.conv.weight"])
    ops.cac_stats(Slice(p2, 64, 64), Slice(p2, 0, 64), pooled, partials)
    ops.cac_gate(B, H, W, partials, w1, b1, w2, b2, chd, pools)
    ops.cac_spatial(pooled, ws, spd)
    g_pre2 = torch.empty((B, 128, H, W), device=dev); g_in2 = torch.zeros((B, 128, H, W), device=dev)
    dw1, db1, dw2, db2, dws = ops.cac_backward(Slice(goc, 0, 64), Slice(goc, 64, 64), Slice(p2, 0, 64), Slice(p2, 64, 64), chd, spd, pooled, pools, w1, b1, w2, ws,
                     Slice(g_pre2, 0, 64), Slice(g_pre2, 64, 64), Slice(g_in2, 0, 64), Slice(g_in2, 64, 64), accumulate_in=False)
    print(f"blk{i}: ch {rel_rmse(chd.cpu(), taps[f'blk{i}.ch']):.2e} sp {rel_rmse(spd.cpu(), taps[f'blk{i}.sp']):.2e} "
          f"g_pre {rel_rmse(g_pre2[:, :64].cpu(), pre.grad):.2e} g_pre_c {rel_rmse(g_pre2[:, 64:].cpu(), pre_c.grad):.2e} "
          f"| torch32-vs-64 g_pre {rel_rmse(pre.grad, taps64[f'blk{i}.pre'].grad):.2e} g_pre_c {rel_rmse(pre_c.grad, taps64[f'blk{i}.pre_c'].grad):.2e}")
    print(f"      dws {rel_rmse(dws.cpu(), p[f'attention_s{i}.spatial.conv.weight'].grad):.2e} (t32v64 {rel_rmse(p[f'attention_s{i}.spatial.conv.weight'].grad, p64[f'attention_s{i}.spatial.conv.weight'].grad):.2e}) "
          f"dw1 {rel_rmse(dw1.cpu(), p[f'attention_c{i}.mlp.1.weight'].grad):.2e} (t32v64 {rel_rmse(p[f'attention_c{i}.mlp.1.weight'].grad, p64[f'attention_c{i}.mlp.1.weight'].grad):.2e}) "
          f"norms g_out {float(g_out.norm()):.2e} g_outc {float(g_outc.norm()):.2e} dws {float(dws.norm()):.2e}")
