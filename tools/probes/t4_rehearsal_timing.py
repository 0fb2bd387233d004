"""Diagnostics of the sporadically slow 4-rank gloo rehearsal (6 s or ~200 s): run it until a slow one shows, with every
rank dumping its Python stacks every 20 s (CODON_BENCH_DUMP_S), and keep that run's full stderr."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
a4 = ["--gpus", "4", "--backend", "gloo", "--scaling", "strong", "--steps", "2", "--warmup", "1", "--batch", "8", "--height", "64",
      "--width", "96"]
os.environ["CODON_BENCH_DUMP_S"] = "20"
for i in range(4):
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + a4, capture_output=True, text=True, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    dt = time.time() - t
    print(f"{dt:7.1f} s rc={r.returncode}", flush=True)
    if dt > 60:
        open(os.path.join(ROOT, "gpurun_out", "r6_t4_slow.err"), "w").write(r.stderr)
        break
