"""One-rank RCCL sanity check on the GPU box: init the nccl (= RCCL) backend, all-reduce the flat gradient buffer of a
GradSync, broadcast parameters.  The multi-GPU runs are the driver's; this only proves the RCCL path loads and runs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from codon_amd import CODONNet
from codon_amd.dist import GradSync
m = CODONNet().to(dev)
gs = GradSync(m)
gs.broadcast_parameters(0)
gs.flat.fill_(1.0)
w = dist.all_reduce(gs.flat, op=dist.ReduceOp.SUM, async_op=True); w.wait()
t = torch.ones(1 << 20, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): dist.all_reduce(gs.flat)
torch.cuda.synchronize()
print("rccl ok: backend", dist.get_backend(), "world", dist.get_world_size(), "flat sum", float(gs.flat.sum()), gs.numel,
      f"all_reduce(7.46 MB, 1 rank) {(time.perf_counter() - t0) / 20 * 1e6:.0f} us")
dist.destroy_process_group()
