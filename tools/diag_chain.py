import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import codon_oracle as orc
from tests.util import load_case, rel_rmse, target_for
from codon_amd import CODONNet
from codon_amd.autograd import _backward_impl
z, variant, sd, x, y = load_case(sys.argv[1] if len(sys.argv) > 1 else "kat0_x4_2x32x24")
tgt = target_for(x)
def run(dt):
    p = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
    taps = {}
    out = orc.forward(p, x.to(dt), y.to(dt), taps)
    for k, v in taps.items():
        if v.requires_grad: v.retain_grad()
    return p, taps, out
p32, t32, o32 = run(torch.float32)
gup = torch.sign(o32.detach() - tgt) / o32.numel()
o32.backward(gup)
p64, t64, o64 = run(torch.float64)
o64.backward(gup.double())
m = CODONNet(); m.load_state_dict(sd); m = m.cuda()
save = {}
with torch.no_grad():
    out = m._forward_impl(x.cuda(), y.cuda(), save)
    dbg = {}
    G = _backward_impl(m, save, x.cuda(), y.cuda(), gup.cuda(), dbg)
for i in (4, 3, 2, 1, 0):
    g = dbg[f"g_oc{i}"].cpu()
    r_d, r_c = t32[f"blk{i}.out"].grad, t32[f"blk{i}.out_c"].grad
    q_d, q_c = t64[f"blk{i}.out"].grad, t64[f"blk{i}.out_c"].grad
    print(f"blk{i}: g_out hip-vs-t32 {rel_rmse(g[:, :64], r_d):.2e} hip-vs-64 {rel_rmse(g[:, :64], q_d):.2e} t32-vs-64 {rel_rmse(r_d, q_d):.2e} | "
          f"g_out_c hip-vs-t32 {rel_rmse(g[:, 64:], r_c):.2e} hip-vs-64 {rel_rmse(g[:, 64:], q_c):.2e} t32-vs-64 {rel_rmse(r_c, q_c):.2e}")
    # forward activations feeding the masks
    S = save[f"blk{i}"]
for k in ("conv4.weight", "conv6.weight", "attention_s2.spatial.conv.weight", "attention_c0.mlp.1.weight", "conv2.weight"):
    print(f"{k:36s} hip-vs-t32 {rel_rmse(G[k].cpu(), p32[k].grad):.2e} hip-vs-64 {rel_rmse(G[k].cpu(), p64[k].grad):.2e} t32-vs-64 {rel_rmse(p32[k].grad, p64[k].grad):.2e}")
print("---- arg-max agreement between the HIP forward and the torch fp32 forward")
for i in range(5):
    pre2 = save[f"blk{i}"]["pre2"].cpu()
    F_h = torch.cat((pre2[:, 64:], pre2[:, :64]), 1)
    F_t = torch.cat((t32[f"blk{i}.pre_c"], t32[f"blk{i}.pre"]), 1).detach()
    B, C, H, W = F_h.shape
    a_h, a_t = F_h.flatten(2).argmax(2), F_t.flatten(2).argmax(2)       # global max-pool arg per (b,c)
    c_h, c_t = F_h.argmax(1), F_t.argmax(1)                              # channel-max arg per pixel
    mm = (a_h != a_t)
    gaps = []
    for b, c in zip(*torch.nonzero(mm, as_tuple=True)):
        v = F_t[b, c].flatten()
        gaps.append(float((v[a_t[b, c]] - v[a_h[b, c]]) / v[a_t[b, c]].abs()))
    mm2 = (c_h != c_t)
    gaps2 = []
    for b, hh, ww in zip(*torch.nonzero(mm2, as_tuple=True)):
        v = F_t[b, :, hh, ww]
        gaps2.append(float((v[c_t[b, hh, ww]] - v[c_h[b, hh, ww]]) / v[c_t[b, hh, ww]].abs()))
    print(f"blk{i}: global-pool arg mismatches {int(mm.sum())}/{mm.numel()} rel gaps {['%.1e' % g for g in gaps[:6]]} ; "
          f"channel-max arg mismatches {int(mm2.sum())}/{mm2.numel()} rel gaps {['%.1e' % g for g in gaps2[:6]]}")
print("---- ReLU mask agreement (HIP saved activations vs torch fp32 recompute), colour + depth streams")
import torch.nn.functional as F
for i in range(5):
    S = save[f"blk{i}"]
    xin = (t32["inputs"], t32["inputs_c"]) if i == 0 else (t32[f"blk{i-1}.out"], t32[f"blk{i-1}.out_c"])
    w = lambda k: sd[k + ".weight"]
    with torch.no_grad():
        st_d = torch.cat((F.relu(F.conv2d(xin[0], w("conv1"), None, 1, 1)), F.relu(F.conv2d(xin[0], w("conv2"), None, 1, 2))), 1)
        st_c = torch.cat((F.relu(F.conv2d(xin[1], w("conv4"), None, 1, 2)), F.relu(F.conv2d(xin[1], w("conv5"), None, 1, 1))), 1)
        r2_d = F.relu(F.conv2d(st_d, w("conv3"), None, 1, 2)); r2_c = F.relu(F.conv2d(st_c, w("conv6"), None, 1, 2))
    for nm, h, t in (("stage", S["stage"], st_d), ("stage_c", S["stage_c"], st_c), ("r2", S["r2"], r2_d), ("r2_c", S["r2_c"], r2_c)):
        h = h.cpu()
        mm = (h > 0) != (t > 0)
        if int(mm.sum()):
            print(f"blk{i} {nm}: {int(mm.sum())} mask flips; |values| there: hip {h[mm].abs().max():.2e} torch {t[mm].abs().max():.2e}")
