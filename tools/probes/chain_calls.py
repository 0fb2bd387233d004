"""Per call of ops.conv_chain1x1 in one fp32 370 x 463 forward: event time and the operands' addresses / channel slices.
chain_calls.py <tree root>"""
import sys
import os

root = sys.argv[1]
sys.path.insert(0, root)
import torch
from codon_amd import CODONNet, ops

torch.manual_seed(0)
m = CODONNet().cuda().eval()
x = torch.rand((1, 1, 370, 463), device="cuda")
y = torch.rand((1, 1, 370, 463), device="cuda")
orig = ops.conv_chain1x1
log = []


def desc(s):
    if s is None:
        return "-"
    if isinstance(s, torch.Tensor):
        return f"{s.data_ptr():#x}"
    return f"{s.buf.data_ptr():#x}[{s.coff}:{s.coff + s.c}/{s.ctotal}]"


def wrapped(*a, **k):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(*a, **k)
    e1.record()
    torch.cuda.synchronize()
    log.append((e0.elapsed_time(e1), [desc(v) for v in a[:4]], {kk: desc(v) for kk, v in k.items() if kk in ("mid", "residual")}))
    return r


with torch.no_grad():
    for _ in range(3):
        m(x, y)
    ops.conv_chain1x1 = wrapped
    m(x, y)
    ops.conv_chain1x1 = orig
print(os.path.basename(os.path.abspath(root)))
for t, a, k in log:
    print(f"{t:7.3f} ms  x {a[0]}  w {a[1]}  wc {a[2]}  out {a[3]}  {k}")
