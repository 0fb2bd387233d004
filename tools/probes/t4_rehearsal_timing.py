import subprocess, sys, time, os
ROOT="/root/repo"
def run(args, cwd):
    t=time.time()
    r=subprocess.run([sys.executable, os.path.join(ROOT,"bench.py")]+args, capture_output=True, text=True, cwd=cwd,
                     env={k:v for k,v in os.environ.items() if k not in ("RANK","WORLD_SIZE","LOCAL_RANK")})
    print(f"{time.time()-t:7.1f} s rc={r.returncode} cwd={cwd} {' '.join(args)[:60]}", flush=True)
    return r
a4=["--gpus","4","--backend","gloo","--scaling","strong","--steps","2","--warmup","1","--batch","8","--height","64","--width","96"]
for i in range(5):
    run(a4, ROOT)
