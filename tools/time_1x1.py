import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codon_amd import ops, _lib as L
from codon_amd.ops import Slice
dev = torch.device("cuda:0"); dt = torch.bfloat16
B, H, W = 32, 480, 640
def run(ci, co, mode, **kw):
    x = torch.randn((B, ci, H, W), device=dev).to(dt)
    w = torch.randn((co, ci, 1, 1), device=dev) * 0.05
    wp = ops.packed_weight(w, mode, dt) if mode == L.PACK_FWD else ops.packed_weight(torch.randn((ci, co, 1, 1), device=dev) * 0.05, mode, dt)
    y = torch.empty((B, co, H, W), device=dev, dtype=dt)
    m = torch.randn((B, co, H, W), device=dev).to(dt)
    args = dict(kw)
    if args.pop("mask", False): args["relu_mask"] = Slice(m)
    if args.pop("res", False): args["residual"] = Slice(m)
    ops.conv2d(Slice(x), wp, Slice(y), 1, **args); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.conv2d(Slice(x), wp, Slice(y), 1, **args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    gb = (ci + co + (co if ("relu_mask" in args or "residual" in args) else 0)) * B * H * W * 2 / 1e9
    print(f"1x1 {ci}->{co} {kw}: {ms:.3f} ms  {gb/ms:.2f} TB/s")
run(128, 64, L.PACK_FWD)
run(128, 64, L.PACK_FWD, res=True)
run(64, 128, L.PACK_DGRAD)
run(64, 128, L.PACK_DGRAD, mask=True)
run(64, 128, L.PACK_DGRAD, accumulate=True)
