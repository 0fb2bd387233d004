"""Where a per-image iteration of the reference's test loop spends its host time (one 463 x 370 RGB guidance + grey depth +
label, fp16): PNG decode, conversion, upload, forward + metrics, download, PNG encode -- then the serial and the pipelined loop."""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
import bench
from codon_amd import CODONNet, infer, io, metrics

dev = torch.device("cuda:0")
r = bench.script_loop_throughput(dev)
print(json.dumps({k: v for k, v in r.items() if k not in ("what", "data", "cpu_oracle")}))
tmp = tempfile.mkdtemp()
from PIL import Image
g = np.random.default_rng(0)
lo = g.random((48, 60, 3))
img = (np.kron(lo, np.ones((8, 8, 1)))[:370, :463] * 200 + g.random((370, 463, 3)) * 55).astype(np.uint8)
Image.fromarray(img, mode="RGB").save(os.path.join(tmp, "c.png"))
Image.fromarray(img[:, :, 0], mode="L").save(os.path.join(tmp, "d.png"))


def t(fn, n=20):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n * 1e3


m = CODONNet().to(dev).half().eval()
print(f"decode RGB->L {t(lambda: io.read_gray(os.path.join(tmp, 'c.png'))):.2f} ms, decode L {t(lambda: io.read_gray(os.path.join(tmp, 'd.png'))):.2f} ms")
p = io.read_gray(os.path.join(tmp, "d.png"))
print(f"convert (numpy) {t(lambda: torch.from_numpy(((np.asarray(p) / 255).astype(np.float32)).astype(np.float16))):.2f} ms, "
      f"convert (torch) {t(lambda: io.to_input(p).to(torch.float16)):.2f} ms")
x = torch.from_numpy(((np.asarray(p) / 255).astype(np.float32)).astype(np.float16))[None, None]
print(f"pin + upload {t(lambda: (x.pin_memory().to(dev, non_blocking=True), torch.cuda.synchronize())):.2f} ms")
xd = x.to(dev)
lab = torch.from_numpy(p.copy()).to(dev)


def fwd():
    with torch.no_grad():
        o = m(xd, xd)
    u = metrics.postprocess_u8(o[0, 0])
    metrics.masked_rmse(lab, u)
    metrics.ssim(lab.float() / 255, u.float() / 255)
    return u


print(f"forward + post + metrics (host-synchronous) {t(fwd):.2f} ms")
u = fwd()
print(f"download {t(lambda: u.cpu()):.2f} ms, encode + write {t(lambda: io.write_gray(os.path.join(tmp, 'o.png'), u.cpu().numpy())):.2f} ms")
