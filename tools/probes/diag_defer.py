import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import codon_oracle as orc
from codon_amd import CODONNet, autograd
sd = orc.he_state("x4", seed=23)
rng = np.random.default_rng(5)
B, H, W = 3, 37, 70
x = torch.from_numpy(rng.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32)).cuda()
y = torch.from_numpy(rng.uniform(0, 1, size=(B, 1, H, W)).astype(np.float32)).cuda()
up = torch.from_numpy(rng.standard_normal(size=(B, 1, H, W)).astype(np.float32)).cuda() / (B * H * W)
for dtype in (None, torch.bfloat16):
    grads = []
    for defer in (False, True, False):
        autograd.DEFER_REDUCE = defer
        m = CODONNet(); m.load_state_dict(sd); m = m.cuda().train()
        if dtype is not None: m.set_compute_dtype(dtype)
        m(x, y).backward(up)
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
    for k in grads[0]:
        a, b, c = grads[0][k].double(), grads[1][k].double(), grads[2][k].double()
        print(dtype, k, "imm-vs-defer %.3e" % float((a - b).norm() / (a.norm() + 1e-30)), "imm-vs-imm %.3e" % float((a - c).norm() / (a.norm() + 1e-30)))
