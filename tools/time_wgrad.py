import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codon_amd import ops
from codon_amd.ops import Slice
dev = torch.device("cuda:0")
dt = {"f32": torch.float32, "bf16": torch.bfloat16}[sys.argv[1] if len(sys.argv) > 1 else "f32"]
B, H, W = int(os.environ.get("B", 32)), 480, 640
cases = [(5, 128, 128), (5, 64, 64), (3, 64, 64), (1, 128, 64)]
if len(sys.argv) > 2:
    cases = [cases[int(sys.argv[2])]]
for (k, ci, co) in cases:
    x = torch.randn((B, ci, H, W), device=dev)
    if os.environ.get("DATA") == "relu":
        x = torch.relu(x)
    x = ops.from_nchw(x, dt)
    g = ops.from_nchw(torch.randn((B, co, H, W), device=dev), dt)
    dw = torch.empty((co, ci, k, k), device=dev)
    n = 3 if k == 5 and ci == 128 else 12
    for _ in range(n):                      # warm-up: the clock settles over some tens of milliseconds
        ops.conv2d_wgrad(Slice(x), Slice(g), dw, k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ops.conv2d_wgrad(Slice(x), Slice(g), dw, k)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * k * k * ci * co * B * H * W
    print(f"wgrad {dt} k{k} {ci}->{co}: {ms:.2f} ms  {fl/ms/1e9:.1f} TFLOP/s")
    del x, g
