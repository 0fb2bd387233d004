"""The driver's contract with bench.py: ONE JSON line on stdout with the agreed keys, at N = 1 and through the
self-launching N > 1 path (2 ranks sharing the one GPU over gloo -- RCCL refuses two ranks on one device; the code path
is the same apart from the backend string)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(args, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_single_gpu_small_shape():
    d = _run(["--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64", "--width", "96"])
    assert REQUIRED <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    rf = d["roofline"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(rf) and 0 < rf["frac"] < 1.05
    cb = d["cpu_baseline"]
    assert {"value", "unit", "cores", "kind", "sample"} <= set(cb) and cb["kind"] == "port" and cb["cores"] >= 1
    fb = d["fwd_bwd"]                                   # the iters/sec (fwd+bwd) half of the BASELINE metric
    assert fb["it_per_s"] > 0 and fb["dtype"] == "bf16" and fb["steps"] >= 3 and fb["rccl_ranks"] == 1
    assert d["rccl_ranks"] == 1


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    d = _run(["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64",
              "--width", "96"])
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
    assert d["fwd_bwd"]["rccl_ranks"] == 2 and d["fwd_bwd"]["global_batch"] == 4
    # whole-job throughput: both ranks' images are counted
    assert abs(d["value"] - 2 * 2 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-6


def test_bench_refuses_without_gpu_and_propagates_child_failure():
    """CPU-checkable half of the launcher: no GPU -> every child exits non-zero -> the parent does too (no JSON line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs an MI355X" in r.stderr and "{" not in r.stdout
