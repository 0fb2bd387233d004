"""Run ON THE GPU BOX: the CAC gate of a block at 32 x 480 x 640 (16-bit path operands): four launches vs codon_cac_tail_fwd."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from codon_amd import ops, _lib as L
dev = torch.device("cuda:0")
B, H, W = int(os.environ.get("B", 32)), int(os.environ.get("H", 480)), int(os.environ.get("W", 640))
nt = ops.cac_fused_tiles(H, W)
r = lambda *s: torch.randn(*s, device=dev)
partials, pc, pd = r(B, nt, 128, 2), r(B, 2, H, W), r(B, 2, H, W)
w1, b1, w2, b2, ws = r(8, 128) * .1, r(8) * .1, r(64, 8) * .3, r(64) * .1, r(1, 2, 5, 5) * .2
ch, sp, po = torch.empty((B, 64), device=dev), torch.empty((B, 1, H, W), device=dev), torch.empty((B, 2, 128), device=dev)
pooled = torch.empty((B, 2, H, W), device=dev)
folded = torch.empty((B, L.CAC_FOLDS, 128, 2), device=dev)
cnt = torch.zeros((B,), dtype=torch.int32, device=dev)
def four():
    ops.cac_fused_finish(B, H, W, partials, pc, pd, folded, pooled)
    ops.cac_gate_folded(B, H, W, folded, w1, b1, w2, b2, ch, po)
    ops.cac_spatial(pooled, ws, sp)
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rep in range(2):
    print(f"{B}x{H}x{W}: four launches {t(four):.1f} us | tail keep {t(lambda: ops.cac_tail(B, H, W, partials, pc, pd, pooled, folded, cnt, w1, b1, w2, b2, ws, ch, sp, po)):.1f} us"
          f" | tail inference {t(lambda: ops.cac_tail(B, H, W, partials, pc, pd, None, folded, cnt, w1, b1, w2, b2, ws, ch, sp, None)):.1f} us")
