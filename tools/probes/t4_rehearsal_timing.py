"""The sporadically slow 4-rank gloo rehearsal on one card (6 s or ~200 s): which arguments matter?  Each variant twice."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
base = ["--gpus", "4", "--backend", "gloo", "--steps", "2", "--warmup", "1"]
variants = {
    "strong b8 64x96": ["--scaling", "strong", "--batch", "8", "--height", "64", "--width", "96"],
    "strong b4 48x64": ["--scaling", "strong", "--batch", "4", "--height", "48", "--width", "64"],
    "weak b2 64x96": ["--batch", "2", "--height", "64", "--width", "96"],
    "strong b8 64x96 no-fwd-bwd": ["--scaling", "strong", "--batch", "8", "--height", "64", "--width", "96", "--no-fwd-bwd"],
    "strong b8 64x96 selfcheck off": ["--scaling", "strong", "--batch", "8", "--height", "64", "--width", "96", "--rccl-selfcheck", "off"],
}
t_all = time.time()
for name, extra in variants.items():
    for i in range(2):
        if time.time() - t_all > 800:
            break
        t = time.time()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + base + extra, capture_output=True, text=True, cwd=ROOT,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
        print(f"{time.time() - t:7.1f} s rc={r.returncode}  {name}", flush=True)
