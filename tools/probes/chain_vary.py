"""Which operand of the slow block call of ops.conv_chain1x1 (fp32 370 x 463 forward) makes it slow?  chain_vary.py <tree root>"""
import sys
import os

root = sys.argv[1]
sys.path.insert(0, root)
import torch
from codon_amd import CODONNet, ops, _lib as L
from codon_amd.ops import Slice

torch.manual_seed(0)
m = CODONNet().cuda().eval()
x = torch.rand((1, 1, 370, 463), device="cuda")
y = torch.rand((1, 1, 370, 463), device="cuda")
orig = ops.conv_chain1x1
calls = []


def wrapped(*a, **k):
    calls.append((a, k))
    return orig(*a, **k)


def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


with torch.no_grad():
    for _ in range(3):
        m(x, y)
    ops.conv_chain1x1 = wrapped
    m(x, y)
    ops.conv_chain1x1 = orig
    print(os.path.basename(os.path.abspath(root)))
    for idx in (0, 1, len(calls) - 1):
        a, k = calls[idx]
        xs, w5, wc, out = a[:4]
        print(f" call {idx}: as recorded (stale data in the buffers) {t(lambda: orig(*a, **k)):.3f} ms")
        st = xs.buf.float().abs()
        print(f"   x: mean |x| {st.mean().item():.3e}  max {st.max().item():.3e}  zeros {(xs.buf == 0).float().mean().item():.3f}"
              f"  subnormal {((st > 0) & (st < 1.1754944e-38)).float().mean().item():.4f}  nan {torch.isnan(xs.buf).any().item()}")
        ws = w5.float().abs()
        print(f"   w5: mean {ws.mean().item():.3e} max {ws.max().item():.3e}   wc: mean {wc.float().abs().mean().item():.3e}")
        keep = xs.buf.clone()
        xs.buf.copy_(torch.relu(torch.randn_like(xs.buf)))
        print(f"   x := relu(randn)            {t(lambda: orig(*a, **k)):.3f} ms")
        xs.buf.zero_()
        print(f"   x := 0                      {t(lambda: orig(*a, **k)):.3f} ms")
        xs.buf.copy_(keep)
        w5r = ops.packed_weight(torch.randn((128, 128, 5, 5), device="cuda") * 0.02, L.PACK_FWD, torch.float32)
        print(f"   w5 := randn * 0.02          {t(lambda: orig(xs, w5r, wc, out, *a[4:], **k)):.3f} ms")
        wcr = ops.packed_weight(torch.randn((64, 128, 1, 1), device="cuda") * 0.1, L.PACK_CHAIN1X1, torch.float32)
        print(f"   wc := randn * 0.1           {t(lambda: orig(xs, w5, wcr, out, *a[4:], **k)):.3f} ms")
        print(f"   both weights random         {t(lambda: orig(xs, w5r, wcr, out, *a[4:], **k)):.3f} ms")
        o2 = torch.empty_like(out.buf)
        print(f"   out := fresh buffer         {t(lambda: orig(xs, w5, wc, Slice(o2, out.coff, out.c), *a[4:], **k)):.3f} ms")
