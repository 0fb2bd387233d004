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
    assert d["rccl_ranks"] == 1 and "grad_equal" not in d          # the self-check is a multi-rank leg


@pytest.mark.gpu
def test_bench_self_launches_two_ranks():
    d = _run(["--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64",
              "--width", "96"])
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["backend"] == "gloo"
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
    assert d["grad_equal"] is True, d.get("rccl_selfcheck")
    assert d["fwd_bwd"]["rccl_ranks"] == 2 and d["fwd_bwd"]["global_batch"] == 4
    # whole-job throughput: both ranks' images are counted
    assert abs(d["value"] - 2 * 2 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-6


@pytest.mark.gpu
def test_bench_survives_an_rccl_that_cannot_form_a_communicator():
    """--backend nccl with two ranks on the ONE GPU of the box: RCCL refuses (duplicate device).  That is this pool's only
    way to exercise the failure path of the 8-GPU run's first RCCL collective: the probe reports it, every collective moves
    to the gloo control group, and the line -- forward throughput, self-check, training leg -- is still printed."""
    env = dict(os.environ, CODON_BENCH_SHARE_GPU="1", CODON_RCCL_TIMEOUT_S="120")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "nccl", "--steps", "2",
                        "--warmup", "1", "--batch", "2", "--height", "64", "--width", "96"], capture_output=True, text=True,
                       timeout=900, env={k: v for k, v in env.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["value"] > 0
    assert d["rccl_probe"]["ok"] is False and d["rccl_probe"]["error"], d["rccl_probe"]
    assert d["backend"] == "gloo" and d["control_backend"] == "gloo"
    assert d["grad_equal"] is True, d.get("rccl_selfcheck")          # the data path itself is sound: on gloo it agrees
    assert d["fwd_bwd"]["rccl_ranks"] == 2


@pytest.mark.gpu
def test_bench_under_the_drivers_torchrun_line():
    """The driver's own launch line for N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N ...`): RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the launcher.
    Two ranks sharing the one GPU, gloo (the RCCL-refuses variant of the same launch is the test above)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend",
                        "gloo", "--steps", "2", "--warmup", "1", "--batch", "2", "--height", "64", "--width", "96",
                        "--no-fwd-bwd"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert REQUIRED <= set(d) and d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["grad_equal"] is True
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and [x["local_rank"] for x in d["ranks"]] == [0, 0]


@pytest.mark.gpu
def test_bench_four_ranks_strong_scaling_and_diagnostics():
    """4 self-launched ranks sharing the one GPU (gloo rehearsal of the 8-GPU run), --scaling strong: 8 images IN TOTAL,
    2 per rank; rank 0's line carries what a mis-bound multi-GPU run would need to be diagnosed from the record alone."""
    # (48 x 64: with 64 x 96 images this rehearsal runs in 6 s or, sporadically -- under pytest always --, in ~190 s: four processes
    # time-slicing one card, every rank waiting in the gradient all-reduce; profiles/r06_rehearsal_4rank_variants.txt)
    d = _run(["--gpus", "4", "--backend", "gloo", "--scaling", "strong", "--steps", "2", "--warmup", "1", "--batch", "8",
              "--height", "48", "--width", "64"])
    assert d["n_gpus"] == 4 and d["scaling"] == "strong" and d["rccl_ranks"] == 4
    assert d["config"]["batch_per_gpu"] == 2 and d["config"]["global_batch"] == 8
    assert abs(d["value"] - 8 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-6
    assert [r["rank"] for r in d["ranks"]] == [0, 1, 2, 3] and all(r["device"] and r["hostname"] for r in d["ranks"])
    assert d["versions"]["hip"] and d["versions"]["torch"]
    fb = d["fwd_bwd"]
    assert fb["scaling"] == "strong" and fb["global_batch"] == 8 and fb["rccl_ranks"] == 4
    assert fb["allreduce_us"] > 0 and fb["allreduce_bytes"] == 1865506 * 4
    assert fb["step_events"]["n"] == fb["steps"] and fb["step_events"]["min_ms"] <= fb["step_events"]["median_ms"]
    assert d["step_events"]["n"] == d["steps"]
    assert fb["roofline"]["launches_timed"] == 13 * fb["steps"] and 0 < fb["roofline"]["frac"] < 1.05
    # 8-GPU first-contact checklist, rehearsed on 4 gloo ranks: the gradient-equality self-check ran
    # over the process group before timing, and every rank's own step times are in the record (a straggler is visible)
    sc = d["rccl_selfcheck"]
    assert d["grad_equal"] is True and sc["first_forward_equal"] is True and sc["ranks"] == 4, sc
    assert set(sc["worst_rel"]) == {"f32", "bf16"} and all(v <= sc["tol"] for v in sc["worst_rel"].values()), sc
    for r in d["ranks"]:
        assert r["fwd_step_ms"]["min_ms"] <= r["fwd_step_ms"]["median_ms"] <= r["fwd_step_ms"]["max_ms"]
        assert r["train_step_ms"]["min_ms"] <= r["train_step_ms"]["median_ms"]


HW_KEYS = {"power_w_p50", "power_w_p95", "sclk_mhz_p50", "power_cap_w", "hwmon_samples", "hwmon_matched_by"}


@pytest.mark.gpu
def test_bench_six_ranks_weak_scaling_rehearsal_of_the_eight_gpu_line():
    """The driver's N = 8 line (`bench.py --gpus 8 --steps K --warmup W`: weak scaling, the self-check, both legs) rehearsed
    with as many ranks as this pool lets share one card: SIX (the box's process guard kills a run with more than six
    processes on the GPU, so the N = 8 case itself cannot be started here; the code path does not depend on N).  Every
    rank's record carries its socket power and shader clock over the timed regions (hwmon), so that a node-level power
    budget on the 8-GPU node reads as eight lower clocks in SCALE_rNN.json rather than as an unexplained efficiency loss."""
    d = _run(["--gpus", "6", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--batch", "1", "--height", "48",
              "--width", "64"], timeout=900)
    assert REQUIRED <= set(d) and d["n_gpus"] == 6 and d["rccl_ranks"] == 6 and d["scaling"] == "weak"
    assert d["config"]["batch_per_gpu"] == 1 and d["config"]["global_batch"] == 6
    assert abs(d["value"] - 6 * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) / d["value"] < 1e-6
    assert [r["rank"] for r in d["ranks"]] == list(range(6))
    sc = d["rccl_selfcheck"]
    assert d["grad_equal"] is True and sc["ranks"] == 6, (sc.get("worst_rel"), sc.get("error"))
    # six is not a power of two: the single-process side sums per-image-mean losses, so both sides see the same 16-bit
    # activation gradients and agree to fp32 summation order in bf16 too (dist.grad_equality_selfcheck)
    assert sc["worst_rel"]["f32"] <= 2e-5 and sc["worst_rel"]["bf16"] <= sc["tol_bf16"] == 2e-5, sc["worst_rel"]
    fb = d["fwd_bwd"]
    assert fb["rccl_ranks"] == 6 and fb["global_batch"] == 6 and fb["scaling"] == "weak" and fb["allreduce_us"] > 0
    for r in d["ranks"]:
        assert HW_KEYS <= set(r) and "train_power_w_p50" in r and "train_sclk_mhz_p50" in r, r
        assert r["fwd_step_ms"]["min_ms"] <= r["fwd_step_ms"]["median_ms"] and r["train_step_ms"] is not None


@pytest.mark.gpu
def test_bench_single_gpu_line_carries_power_and_clock_and_the_profile_hash():
    d = _run(["--steps", "3", "--warmup", "1", "--batch", "4", "--height", "96", "--width", "128", "--no-cpu-baseline"])
    r0 = d["ranks"][0]
    assert HW_KEYS <= set(r0) and "train_power_w_p50" in r0
    if r0["hwmon_samples"]:                  # sysfs visible on this box: plausible numbers for an MI355X under load
        assert 50 < r0["power_w_p50"] < 1600 and 300 < r0["sclk_mhz_p50"] < 3000, r0
    rf = d["roofline"]
    assert "traffic_from_hash" in rf and len(rf["lib_source_hash"]) == 32
    assert "traffic_from_hash" in d["fwd_bwd"]["roofline"]


@pytest.mark.gpu
def test_bench_retries_a_taken_rendezvous_port_and_times_out(capfd):
    """self_launch: a port that is taken between bind-and-close and the children's bind is retried on another one; a
    wall-clock limit terminates the children (exact PIDs) and returns non-zero."""
    import socket
    sys.path.insert(0, ROOT)
    import bench
    busy = socket.socket()
    busy.bind(("127.0.0.1", 0))
    busy.listen(1)
    real, calls = bench._free_port, []

    def fake():
        calls.append(1)
        return busy.getsockname()[1] if len(calls) == 1 else real()

    bench._free_port = fake
    try:
        args = ["--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--batch", "1", "--height", "32",
                "--width", "32", "--no-fwd-bwd"]
        env = {k: os.environ.pop(k) for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK") if k in os.environ}
        rc = bench.self_launch(2, args)
        out = capfd.readouterr()
        assert rc == 0 and len(calls) >= 2, out.err[-3000:]
        assert "retrying on another port" in out.err and out.out.count('"metric"') == 1
        rc = bench.self_launch(2, args, timeout_s=0.5)
        out = capfd.readouterr()
        assert rc == 124 and "terminated" in out.err
        os.environ.update(env)
    finally:
        bench._free_port = real
        busy.close()


def test_bench_refuses_without_gpu_and_propagates_child_failure():
    """CPU-checkable half of the launcher: no GPU -> every child exits non-zero -> the parent does too (no JSON line)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "needs an MI355X" in r.stderr and "{" not in r.stdout
