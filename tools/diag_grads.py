import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import load_case, rel_rmse, target_for
from codon_amd import CODONNet, CODONNet16
for name in sys.argv[1:] or ["kat0_x4_2x32x24", "he0_x4_2x24x20_taps"]:
    z, variant, sd, x, y = load_case(name)
    m = (CODONNet16 if variant == "x16" else CODONNet)(); m.load_state_dict(sd); m = m.cuda()
    out = m(x.cuda(), y.cuda())
    tgt = target_for(x)
    ref_out = torch.from_numpy(z["out"])
    out.backward((torch.sign(ref_out - tgt) / ref_out.numel()).cuda())
    print(name)
    for k, p in m.named_parameters():
        if p.grad is None: continue
        stride = int(z["gradstride." + k])
        got = p.grad.flatten()[::stride].cpu()
        print(f"  {k:40s} rel_rmse {rel_rmse(got, z['grad.'+k]):.3e}  norm got {float(p.grad.norm()):.6e} ref {float(z['gradnorm.'+k]):.6e}")
