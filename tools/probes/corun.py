"""Run ON THE GPU BOX: does an HBM-bound kernel hide behind a power-limited MFMA kernel when both run at once?
wgrad 5x5 128->128 (bf16, matrix-bound, at the socket's power cap) and ew_sum_mask (4 x 64 channels in, 64 out: streaming)
back to back on one stream vs on two streams."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from codon_amd import ops
from codon_amd.ops import Slice
dev = torch.device("cuda:0")
dt = torch.bfloat16
B, H, W = 32, 480, 640
x = ops.from_nchw(torch.relu(torch.randn((B, 128, H, W), device=dev)), dt)
g = ops.from_nchw(torch.randn((B, 128, H, W), device=dev), dt)
dw = torch.empty((128, 128, 5, 5), device=dev)
srcs = [ops.from_nchw(torch.randn((B, 64, H, W), device=dev), dt) for _ in range(4)]
dst = ops.new_act(B, 64, H, W, dt, dev)
NE = int(os.environ.get("NE", 4))      # streaming launches per wgrad launch
def mfma():
    ops.conv2d_wgrad(Slice(x), Slice(g), dw, 5)
def hbm():
    for _ in range(NE):
        ops.ew_sum_mask(Slice(dst), [Slice(t) for t in srcs], mask=Slice(srcs[0]))
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def timed(fn, n=8):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def serial():
    mfma(); hbm()
def corun():
    cur = torch.cuda.current_stream(dev)
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        mfma()
    with torch.cuda.stream(s2):
        hbm()
    cur.wait_stream(s1); cur.wait_stream(s2)
for rep in range(2):
    a, b = timed(mfma), timed(hbm)
    c, d = timed(serial), timed(corun)
    print(f"wgrad alone {a:.2f} ms, {NE} x ew_sum_mask alone {b:.2f} ms, back to back {c:.2f} ms, on two streams {d:.2f} ms")
