"""Full-size (32 x 480 x 640) bf16 training step, three times on the same inputs: every parameter gradient must be bit-identical
between runs (fixed-order reductions everywhere; a race in the hand-synchronised kernels -- LDS-DMA + asm reads with counted
waits, emitting gated convs, the fused 1x1 backward -- would show up here as a mismatch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import codon_amd
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = codon_amd.CODONNet().to(dev)
net.set_compute_dtype(torch.bfloat16)
net.train()
B, H, W = int(os.environ.get("B", 32)), 480, 640
x, y, gy = torch.rand((B, 1, H, W), device=dev), torch.rand((B, 1, H, W), device=dev), torch.randn((B, 1, H, W), device=dev)
ref = None
for it in range(3):
    net.zero_grad(set_to_none=True)
    out = net(x, y)
    out.backward(gy)
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    g["__out__"] = out.detach().clone()
    if ref is None:
        ref = g
        assert all(torch.isfinite(v).all() for v in g.values()), "non-finite gradient"
    else:
        bad = [n for n in ref if not torch.equal(ref[n], g[n])]
        assert not bad, f"run {it}: {len(bad)} tensors differ: {bad[:5]}"
    print(f"run {it}: {len(g)} tensors, |grad| sum {sum(float(v.abs().sum()) for v in g.values()):.6e}", flush=True)
print("deterministic")
