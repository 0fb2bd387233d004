import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codon_amd import ops
from codon_amd.ops import Slice
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16x3": torch.float32}[mode]
split = mode == "f16x3"
from codon_amd import _lib as L
B, H, W = int(os.environ.get("B", 32)), 480, 640
cases = [(5, 128, 128), (5, 64, 64), (3, 64, 64), (3, 128, 64), (1, 128, 64)]
if len(sys.argv) > 2:
    cases = [cases[int(sys.argv[2])]]
for (k, ci, co) in cases:
    x = torch.randn((B, ci, H, W), device=dev)
    if os.environ.get("DATA") == "relu":      # post-ReLU-like activations clock higher than dense random ones
        x = torch.relu(x)
    x = ops.from_nchw(x, dt)
    w = torch.randn((co, ci, k, k), device=dev) * 0.05
    if split and k == 1:
        continue
    wp = ops.packed_weight(w, L.PACK_FWD_F16X3 if split else L.PACK_FWD, dtype=dt)
    y = ops.new_act(B, co, H, W, dt, dev)
    ops.conv2d(Slice(x), wp, Slice(y), k, relu=True, f16x3=split)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d(Slice(x), wp, Slice(y), k, relu=True, f16x3=split)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * k * k * ci * co * B * H * W
    print(f"conv {dt} k{k} {ci}->{co}: {ms:.3f} ms  {fl/ms/1e9:.1f} TFLOP/s")
    del x, y
