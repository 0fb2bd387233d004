"""Run ON THE GPU BOX: bf16 conv5x5 64->64 at 32 x 480 x 640 -- plain, gated, gated + emitting (CODON_AMD_LIB selects the build)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice
dev = torch.device("cuda:0")
dt = torch.bfloat16
B, H, W = 32, 480, 640
torch.manual_seed(0)
pre = ops.from_nchw(torch.randn((B, 64, H, W), device=dev), dt)
inp = ops.from_nchw(torch.relu(torch.randn((B, 64, H, W), device=dev)), dt)
xpl = ops.from_nchw(torch.relu(torch.randn((B, 64, H, W), device=dev)), dt)
ch, sp = torch.rand((B, 64), device=dev), torch.rand((B, 1, H, W), device=dev)
w = ops.packed_weight(torch.randn((64, 64, 5, 5), device=dev) * 0.03, L.PACK_FWD, dt)
y, em = ops.new_act(B, 64, H, W, dt, dev), ops.new_act(B, 64, H, W, dt, dev)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rep in range(2):
    a = t(lambda: ops.conv2d(Slice(xpl), w, Slice(y), 5, relu=True))
    b = t(lambda: ops.conv2d_gated(Slice(pre), Slice(inp), ch, sp, w, Slice(y), 5, relu=True))
    c = t(lambda: ops.conv2d_gated(Slice(pre), Slice(inp), ch, sp, w, Slice(y), 5, relu=True, emit=Slice(em)))
    print(f"{os.environ.get('CODON_AMD_LIB', 'product')[-12:]}: plain {a:.3f}  gated {b:.3f}  gated+emit {c:.3f} ms")
