"""GPU diagnostic behind tests/test_gpu_backward.py::test_bf16_gradients_vs_reference_bf16_autograd: per tensor, the HIP error vs
the reference's float64 gradient in fp32 / bf16-compute / whole-module-bf16, pooled (RMS) over the input variants of a
bf16grad_* fixture, beside the reference module's own bf16 error (tools/make_golden_r4.py) and its spread over the variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.util import BF16_GRAD_CASES, bf16grad_inputs, load_case
from codon_amd import CODONNet, CODONNet16

for name in BF16_GRAD_CASES:
    z, variant, sd, _, _ = load_case(name)
    B, H, W = (int(v) for v in z["shape"]); nv = int(z["nv"])
    res = {}
    for mode in ("f32", "bf16c", "bf16m"):
        m = (CODONNet16 if variant == "x16" else CODONNet)()
        m.load_state_dict(sd); m = m.cuda().train()
        if mode == "bf16c": m.set_compute_dtype(torch.bfloat16)
        if mode == "bf16m": m = m.bfloat16()
        for v in range(nv):
            x, y = bf16grad_inputs(v, B, H, W)
            up = torch.from_numpy(z[f"v{v}.upstream"]).cuda()
            m.zero_grad(set_to_none=True)
            out = m(x.cuda().bfloat16(), y.cuda().bfloat16()) if mode == "bf16m" else m(x.cuda(), y.cuda())
            out.backward(up.to(out.dtype))
            for k, p in m.named_parameters():
                if p.grad is None: continue
                s = int(z["stride." + k]); g64 = z[f"v{v}.g64.{k}"].astype(np.float64)
                got = p.grad.detach().float().flatten()[::s].cpu().double().numpy()
                res.setdefault(k, {}).setdefault(mode, []).append(float(((got - g64) ** 2).sum() / (g64 ** 2).sum()))
    print(f"== {name}: tensor | reference bf16 err min/RMS/max over {nv} variants | HIP RMS err: f32, bf16-compute (ratio), module-bf16 (ratio)")
    for k, r in res.items():
        er = [float(z[f"v{v}.err_sub.{k}"]) for v in range(nv)]
        rr = float(np.sqrt(np.mean(np.square(er))))
        h = {mo: float(np.sqrt(np.mean(r[mo]))) for mo in r}
        print(f"{k:36s} ref {min(er):.2e} {rr:.2e} {max(er):.2e} | f32 {h['f32']:.1e}  bf16c {h['bf16c']:.2e} ({h['bf16c']/rr:4.2f})  bf16m {h['bf16m']:.2e} ({h['bf16m']/rr:4.2f})")
