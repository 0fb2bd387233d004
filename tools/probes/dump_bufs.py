"""Run ON THE GPU BOX: addresses of the activation buffers of a one-image fp32 forward (TREE selects the source tree)."""
import sys, os, torch
sys.path.insert(0, os.environ["TREE"])
from codon_amd import CODONNet, ops
torch.manual_seed(0)
m = CODONNet().cuda().eval()
H, W = 370, 463
x = torch.rand((1, 1, H, W), device="cuda"); y = torch.rand((1, 1, H, W), device="cuda")
seen = []
orig = ops.new_act
def spy(B, c, H_, W_, dt, dev):
    t = orig(B, c, H_, W_, dt, dev)
    seen.append((c, t.data_ptr()))
    return t
ops.new_act = spy
with torch.no_grad():
    for _ in range(3):
        seen.clear()
        m(x, y)
torch.cuda.synchronize()
print(os.environ["TREE"][-8:], " ".join(f"{c}ch@{p:#x}(mod2M={p % (1 << 21):#x})" for c, p in seen))
