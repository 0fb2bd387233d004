import sys, os, torch
sys.path.insert(0, os.environ["TREE"])
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice
dev = torch.device("cuda:0")
B, H, W = 1, 370, 463
torch.manual_seed(0)
x = torch.relu(torch.randn((B, 128, H, W), device=dev))
w5 = ops.packed_weight(torch.randn((128, 128, 5, 5), device=dev) * 0.02, L.PACK_FWD, torch.float32)
w1 = ops.packed_weight(torch.randn((64, 128, 1, 1), device=dev) * 0.1, L.PACK_CHAIN1X1, torch.float32)
o = torch.empty((B, 128, H, W), device=dev)
def t(fn, n=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print(os.environ["TREE"][-8:], "chain1x1 370x463: %.3f ms" % t(lambda: ops.conv_chain1x1(Slice(x), w5, w1, Slice(o, 64, 64))),
      " conv5x5-64: %.3f ms" % t(lambda: ops.conv2d(Slice(x, 0, 64), ops.packed_weight(torch.ones((64,64,5,5), device=dev)*0.01, L.PACK_FWD, torch.float32), Slice(o, 0, 64), 5, relu=True)))
