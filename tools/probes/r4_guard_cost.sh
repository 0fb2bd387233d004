cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4a
python - > gpurun_out/r4a/guard_cost.txt 2>&1 <<'PY'
import os, time, torch, subprocess, sys
code = """
import time, torch
from codon_amd import CODONNet
from codon_amd.graph import GraphedCODON
torch.manual_seed(0)
m = CODONNet().cuda().eval()
x = torch.rand((1,1,128,128), device='cuda'); y = torch.rand((1,1,128,128), device='cuda')
with torch.no_grad():
    gm = GraphedCODON(m, x, y)
    for name, fn in (('eager', lambda: m(x,y)), ('graph', lambda: gm(x,y))):
        best = 1e9
        for rep in range(5):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0=time.perf_counter()
            for _ in range(50): fn()
            torch.cuda.synchronize(); best=min(best,(time.perf_counter()-t0)/50*1e3)
        print(name, '%.4f ms' % best)
"""
for g in ("1", "0", "1", "0"):
    env = dict(os.environ, CODON_WEIGHT_GUARD=g)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("CODON_WEIGHT_GUARD=" + g, r.stdout.replace("\n", "  "), r.stderr[-300:])
PY
cat gpurun_out/r4a/guard_cost.txt
