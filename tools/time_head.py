"""A/B of the head stencil's band height (build with `make EXTRA=-DCODON_TUNE`): ms and algorithmic TB/s at C2 shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codon_amd import ops
from codon_amd.ops import Slice
B, H, W = 32, 480, 640
w = torch.randn(1, 64, 3, 3, device="cuda") * 0.1
res = torch.rand(B, 1, H, W, device="cuda")
y = torch.empty_like(res)
for dt, bands in ((torch.float32, ["204", "208", "216", "402", "404", "408", "416"]), (torch.bfloat16, ["204", "208", "216", "402", "404", "408", "416", "802", "804", "808"])):
    x = torch.randn(B, 64, H, W, device="cuda").to(dt)
    ref = None
    for band in bands:
        os.environ["CODON_HEAD_BAND"] = band
        ops.head(Slice(x), w, res, y)
        if ref is None:
            ref = y.clone()
        assert torch.equal(ref, y), band
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.head(Slice(x), w, res, y)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        alg = B * H * W * (64 * x.element_size() + 8)
        print(f"head {dt} band {band}: {ms:.3f} ms  {alg / ms / 1e9:.2f} TB/s algorithmic")
