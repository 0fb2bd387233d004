"""Single-image latency (the reference script's use case: one image per call, test.py:125)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from codon_amd import CODONNet
from codon_amd.graph import GraphedCODON
torch.manual_seed(0)
for (B, H, W) in [(1, 128, 128), (1, 370, 463), (1, 480, 640)]:
    x, y = torch.rand(B, 1, H, W, device="cuda"), torch.rand(B, 1, H, W, device="cuda")
    for label, mk in (("fp32 exact", lambda: CODONNet().cuda().eval()),
                      ("fp32 f16x3 (opt-in)", lambda: CODONNet().cuda().eval().set_conv_precision("f16x3")),
                      ("fp16 (.half(), as test.py)", lambda: CODONNet().cuda().half().eval()),
                      ("bf16", lambda: CODONNet().cuda().eval().set_compute_dtype(torch.bfloat16))):
        m = mk()
        xi, yi = (x.half(), y.half()) if "half" in label else (x, y)
        with torch.no_grad():
            for _ in range(3): m(xi, yi)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20): m(xi, yi)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print(f"{B}x{H}x{W} {label:28s}: {dt*1e3:7.3f} ms/forward")
    gm = GraphedCODON(CODONNet().cuda().eval(), x, y)
    with torch.no_grad():
        for _ in range(3): gm(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): gm(x, y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"{B}x{H}x{W} {'fp32 exact, hipGraph replay':28s}: {dt*1e3:7.3f} ms/forward")
