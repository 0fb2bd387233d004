#!/usr/bin/env python3
"""bench.py -- CODON x4 forward on MI355X: HR depth maps/s at batch 32/GPU, 480x640, fp32
(BASELINE.json configs[1]).  One "step" = one CODONNet forward over one synthetic batch.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

With --gpus N > 1 and no launcher environment (WORLD_SIZE unset) this script starts the N ranks itself: the parent
makes no GPU call, spawns N fresh `python bench.py` children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
(rendezvous on 127.0.0.1), forwards rank 0's JSON line and exits non-zero if any child fails.

The BASELINE metric has two halves -- "HR depth maps/sec (fwd) + iters/sec (fwd+bwd)": the default line's `value` is
the exact-fp32 forward (configs[1]); the same line carries `fwd_bwd`: a short training leg at configs[2]'s per-GPU
shape (x4, batch 32/GPU, 480x640, bf16 activations, fp32 accumulate + master weights, L1+SSIM loss, one RCCL
all-reduce of the flat gradient per step when N > 1, Adam).

Images are independent units (no op mixes samples), so ranks shard the batch with NO data-path
collective in forward ("scaling": "weak": 32 images per GPU whatever N; `--scaling strong`: 32 images in
total, split over the ranks -- SURVEY.md 8e -- with the label in "scaling").  Rank 0 prints ONE JSON
line.  `roofline` is measured live for the dominant kernel (the 5x5 128->128 fp32 MFMA conv,
71.7 % of the FLOPs) with HIP events recorded on the launch stream around each of its launches
inside the timed region.  `cpu_baseline` times the CPU oracle (a port of the reference's
eager-PyTorch path, validated against the imported reference) on a bounded sample, rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

FLOP_PER_PIXEL_FWD = 14_856_052          # 2 x 7 428 026 MAC
ALG_ELEMS_PER_PIXEL = 12_706             # activation elements moved per output pixel
CONV5_128_MAC_PER_PIXEL = 409_600        # 5*5*128*128
PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2500.0           # MI355X_MICROARCH.md: dense bf16 MFMA (no sparsity)
PEAK_HBM_GBS = 8000.0


def synth_inputs(B, H, W, scale, seed, dev):
    """y = uniform 8-bit grey / 255; x = smooth LR depth field brought to HxW by a
    bicubic x`scale` upsample, clipped to [0,1] (the reference's inputs are pre-upsampled offline,
    CODON_X4/test.py:70-77,116-123)."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    y = torch.randint(0, 256, (B, 1, H, W), generator=g, device=dev).float() / 255.0
    lr = torch.rand((B, 1, H // scale + 2, W // scale + 2), generator=g, device=dev)
    lr = torch.nn.functional.avg_pool2d(lr, 3, 1)                    # smooth: (H/s, W/s)
    from codon_amd.upsample import bicubic_upsample
    x = bicubic_upsample(lr.contiguous(), scale).clamp_(0.0, 1.0)
    return x.contiguous(), y.contiguous()


# profiled workloads other than 32 x 480 x 640 (round 6: BASELINE configs[3] and [4]): (B, H, W) -> PMC table pattern
PMC_SHAPES = {(16, 960, 1280): "r*_x8_fwd_b16_960x1280_pmc.json", (8, 1920, 2560): "r*_x16_bf16_fwd_b8_1920x2560_pmc.json"}


def pmc_pattern(B, H, W, default):
    return default if (B, H, W) == (32, 480, 640) else PMC_SHAPES.get((B, H, W))


def pmc_traffic(kernel_prefix, B, H, W, pattern="r*_fwd_b32_480x640_pmc*.json"):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/: FETCH_SIZE x2-corrected +
    WRITE_SIZE, separate passes of the same bench command).  Counters cannot be read from inside this process, so this
    is the latest committed measurement for this exact workload, or None."""
    pattern = pmc_pattern(B, H, W, pattern)
    if pattern is None:
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        if "_before_" in os.path.basename(path):
            continue
        try:
            ks = json.load(open(path))["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for k, v in ks.items():
            if k.replace("codon::", "").startswith(kernel_prefix.replace("codon::", "")):
                return v["hbm_bytes_per_launch"]
    return None


def pmc_traffic_hash(pattern="r*_fwd_b32_480x640_pmc*.json"):
    """(library source hash the newest committed PMC table of `pattern` was taken from, hash of the library loaded now):
    printed beside roofline.traffic so that a ratio measured on an older build of the kernels is visible from the line."""
    import glob
    from codon_amd import _lib
    now = _lib.build_info()["source_hash_built"]
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)), reverse=True):
        if "_before_" in os.path.basename(path):
            continue
        try:
            return json.load(open(path)).get("lib_source_hash"), now
        except (OSError, ValueError):
            continue
    return None, now


class HwmonSampler:
    """Socket power and shader clock of THIS rank's card during a timed region (hwmon sysfs: power1_average / power1_input,
    freq1_input, power1_cap), sampled every 50 ms by a host thread that never touches the GPU.  A node-level power budget
    -- 8 sockets that each sit at their 1400 W cap when alone -- shows up in the rank records as lower clocks instead of as
    an unexplained loss of weak-scaling efficiency.  The card is found by PCI address (torch's pci_bus_id against the sysfs
    device link); if that fails (no sysfs, a container that hides it) the fields are null -- except with ONE rank, where the
    card that draws the most power during the region is taken (the box shows every card of the host)."""

    def __init__(self, dev):
        import glob
        self.hws = [h for h in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")) if glob.glob(h + "/power1_*")]
        self.mine = None
        try:
            pr = torch.cuda.get_device_properties(dev)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
            for h in self.hws:
                if os.path.basename(os.path.realpath(os.path.join(h, "..", ".."))).startswith(want):
                    self.mine = h
        except Exception:                      # noqa: BLE001 -- telemetry must never fail the bench
            pass
        self.rows = {h: [] for h in self.hws}
        self._stop = None
        self._th = None

    @staticmethod
    def _rd(path):
        try:
            return float(open(path).read())
        except Exception:                      # noqa: BLE001
            return float("nan")

    def start(self):
        import threading
        if not self.hws:
            return self
        self._stop = threading.Event()
        cards = [self.mine] if self.mine else self.hws

        def run():
            while not self._stop.is_set():
                for h in cards:
                    pw = self._rd(h + "/power1_average")
                    if pw != pw:
                        pw = self._rd(h + "/power1_input")
                    self.rows[h].append((pw / 1e6, self._rd(h + "/freq1_input") / 1e6))
                self._stop.wait(0.05)

        self._th = threading.Thread(target=run, name="hwmon", daemon=True)
        self._th.start()
        return self

    def stop(self, single_rank=True):
        out = {"power_w_p50": None, "power_w_p95": None, "sclk_mhz_p50": None, "power_cap_w": None, "hwmon_samples": 0,
               "hwmon_matched_by": None}
        if self._th is None:
            return out
        self._stop.set()
        self._th.join(1.0)
        h = self.mine
        how = "pci address"
        if h is None and single_rank:
            mean = {k: (sum(r[0] for r in v if r[0] == r[0]) / max(1, len(v))) for k, v in self.rows.items()}
            h = max(mean, key=mean.get) if mean else None
            how = "highest draw (one rank)"
        if h is None or not self.rows[h]:
            return out
        pw = sorted(r[0] for r in self.rows[h] if r[0] == r[0])
        fq = sorted(r[1] for r in self.rows[h] if r[1] == r[1])
        cap = self._rd(h + "/power1_cap") / 1e6
        if pw:
            out.update(power_w_p50=pw[len(pw) // 2], power_w_p95=pw[int(len(pw) * 0.95)], hwmon_samples=len(pw))
        if fq:
            out.update(sclk_mhz_p50=fq[len(fq) // 2])
        out.update(power_cap_w=cap if cap == cap else None, hwmon_matched_by=how)
        return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def step_stats(events):
    """median / min of per-step HIP-event pairs recorded on the launch stream."""
    ms = sorted(s.elapsed_time(e) for s, e in events)
    if not ms:
        return None
    return {"median_ms": ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2]),
            "min_ms": ms[0], "max_ms": ms[-1], "n": len(ms)}


def cpu_baseline(H, W):
    """Oracle forward on the host cores for ONE image of the workload (bounded: ~10-30 s)."""
    from oracle import codon_oracle as orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the GPU box exposes every host core but grants a 16-core share per GPU; oversubscribing
    # oneDNN with 256 threads runs ~3x slower than 16
    n = int(os.environ.get("CODON_CPU_THREADS", "0")) or min(avail, 16)
    torch.set_num_threads(n)
    sd = orc.he_state("x4", seed=0)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((1, 1, H, W), generator=g)
    y = torch.rand((1, 1, H, W), generator=g)
    with torch.no_grad():
        # BASELINE.json configs[0] (the reference's own CPU-runnable case, 1x128x128): 1 warm-up + 5 runs
        x1, y1 = x[:, :, :128, :128].contiguous(), y[:, :, :128, :128].contiguous()
        orc.forward(sd, x1, y1)
        c1 = []
        for _ in range(5):
            t0 = time.perf_counter()
            orc.forward(sd, x1, y1)
            c1.append(time.perf_counter() - t0)
        full = []
        orc.forward(sd, x, y)                    # 1 warm-up + 5 runs, ~6 s each on 16 threads: ~35 s
        for _ in range(5):
            t0 = time.perf_counter()
            orc.forward(sd, x, y)
            full.append(time.perf_counter() - t0)
    c1.sort()
    full.sort()
    dt = full[2]
    return {"value": 1.0 / dt, "unit": "maps/s", "cores": torch.get_num_threads(), "cpu_model": cpu_model(),
            "host_cores_visible": avail, "kind": "port",
            "sample": f"1 image (1x1x{H}x{W} pair) of the batch, 1 warm-up + 5 oracle forwards: median {dt:.1f} s, min {full[0]:.1f} s",
            "mpx_per_s": H * W / dt / 1e6,
            "config0_1x128x128": {"min_s": c1[0], "median_s": c1[2], "runs": 5}}


def _latency_ms(fn, dev, warm=20, n=50):
    """Steady-state milliseconds per call: `n` calls after `warm` warm-up calls, device synchronised on both sides."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e3


def config0_gpu_latency(dev):
    """BASELINE.json configs[0] (1 x 128 x 128 pair, fp32) on the GPU, beside the CPU number for the same case: the
    reference script's own use (one image per call, test.py:125).  ~95 kernel launches per forward: eager is
    launch-bound, so the hipGraph replay (codon_amd.graph.GraphedCODON) is reported too."""
    from codon_amd import CODONNet
    from codon_amd.graph import GraphedCODON
    torch.manual_seed(0)
    m = CODONNet().to(dev).eval()
    x, y = torch.rand((1, 1, 128, 128), device=dev), torch.rand((1, 1, 128, 128), device=dev)
    out = {}
    with torch.no_grad():
        gm = GraphedCODON(m, x, y)
        for name, fn in (("eager_ms", lambda: m(x, y)), ("hipgraph_replay_ms", lambda: gm(x, y))):
            out[name] = _latency_ms(fn, dev)
    out["workload"] = "CODON x4 forward, 1 x 128 x 128 pair, fp32 exact (BASELINE.json configs[0]) on 1 MI355X"
    out["protocol"] = "steady state: mean of 50 calls after 20 warm-up calls (the first ~20 calls after an idle gap run 3 % slower: clock ramp, tools/probes/b1_warm.py)"
    return out


def script_pattern_latency(dev):
    """The reference script's real use (CODON_X4/test.py:52,116-125): ONE image per call, model.cuda().half(), at the
    sizes of the shipped Middlebury samples (370x463, 375x450, 247x343) -- eager and as a hipGraph replay, in fp16 and
    in fp32.  Milliseconds per forward, steady state (mean of 50 after 20 warm-up calls)."""
    from codon_amd import CODONNet
    from codon_amd.graph import GraphedCODON
    torch.manual_seed(0)
    res = {"what": "one (1,1,H,W) pair per call, as the reference's test loop does; ms per forward"}
    for dt_name, prep in (("fp16", lambda m: m.half()), ("fp32", lambda m: m)):
        m = prep(CODONNet().to(dev)).eval()
        dtype = torch.float16 if dt_name == "fp16" else torch.float32
        for (H, W) in ((370, 463), (375, 450), (247, 343)):
            x = torch.rand((1, 1, H, W), device=dev).to(dtype)
            y = torch.rand((1, 1, H, W), device=dev).to(dtype)
            with torch.no_grad():
                gm = GraphedCODON(m, x, y)
                for name, fn in (("eager", lambda: m(x, y)), ("hipgraph", lambda: gm(x, y))):
                    res[f"{dt_name}_{H}x{W}_{name}_ms"] = _latency_ms(fn, dev)
            del gm
        del m
    return res


# the ten image pairs the reference ships for x4 (CODON_X4/input_depth, input_color, input_label: Middlebury crops) -- width x
# height as PIL reports them, and whether the guidance PNG is RGB (eight of them) or already grey
SHIPPED_X4 = [("Art", 463, 370, True), ("Books", 463, 370, True), ("Cones", 450, 375, True), ("Dolls", 463, 370, True),
              ("Laundry", 447, 370, True), ("Moebius", 463, 370, True), ("Reindeer", 447, 370, True), ("Rocks", 425, 370, True),
              ("Teddy", 450, 375, False), ("Tsukuba", 343, 247, False)]


def script_loop_throughput(dev):
    """the reference script's WHOLE per-image loop (CODON_X4/test.py:109-145: read two PNGs, grey, /255, upload,
    .half() forward, clip * 255 -> uint8, download, write PNG, masked RMSE + SSIM against the label) over ten image pairs of the
    shipped x4 set's sizes -- synthetic smooth content written to a scratch directory (the shipped files do not travel), random
    init (the weights are not shipped).  codon_amd.infer runs it as a pipeline (reader thread + side-stream uploads | forward +
    metrics | writer thread); the reference-style serial loop is timed beside it (byte-identical outputs:
    tests/test_extras.py::test_infer_cli_end_to_end)."""
    import shutil
    import tempfile
    import numpy as np
    from PIL import Image
    from codon_amd import CODONNet, infer
    tmp = tempfile.mkdtemp(prefix="codon_loop_")
    try:
        g = np.random.default_rng(0)
        for d in ("depth", "color", "label", "out_s", "out_p"):
            os.makedirs(os.path.join(tmp, d))

        def smooth(h, w, c):
            lo = g.random((h // 8 + 2, w // 8 + 2, c))
            img = np.kron(lo, np.ones((8, 8, 1)))[:h, :w] * 200 + g.random((h, w, c)) * 55
            return img.astype(np.uint8)

        for name, w, h, rgb in SHIPPED_X4:
            Image.fromarray(smooth(h, w, 1)[:, :, 0], mode="L").save(os.path.join(tmp, "depth", name + ".png"))
            Image.fromarray(smooth(h, w, 1)[:, :, 0], mode="L").save(os.path.join(tmp, "label", name + ".png"))
            col = smooth(h, w, 3 if rgb else 1)
            Image.fromarray(col if rgb else col[:, :, 0], mode="RGB" if rgb else "L").save(os.path.join(tmp, "color", name + ".png"))
        torch.manual_seed(0)
        m = CODONNet().to(dev).half().eval()
        kw = dict(input_depth=os.path.join(tmp, "depth"), input_color=os.path.join(tmp, "color"), label=os.path.join(tmp, "label"),
                  emit=lambda s_: None)
        res = {"what": "the reference script's per-image loop (test.py:109-145) over 10 PNG pairs of the shipped x4 sizes, "
                       "model.cuda().half(), outputs written, RMSE + SSIM vs the label; whole loop wall time incl. PNG "
                       "decode / encode on the host; second pass of two (first pass warms allocator and page cache)",
               "images": len(SHIPPED_X4), "data": "synthetic PNGs at the shipped sizes (8 RGB + 2 grey guidance images)"}
        for tag, pipe, od in (("serial", False, "out_s"), ("pipelined", True, "out_p")):
            best = None
            for _ in range(2):
                r = infer.run_loop(m, dev, torch.float16, out_dir=os.path.join(tmp, od), pipelined=pipe, **kw)
                best = r
            res[f"{tag}_images_per_s"] = best["images_per_s"]
            res[f"{tag}_ms_per_image"] = 1e3 * best["seconds"] / best["n"]
        res["script_loop_images_per_s"] = res["pipelined_images_per_s"]
        same = all(open(os.path.join(tmp, "out_s", n + ".png"), "rb").read() == open(os.path.join(tmp, "out_p", n + ".png"), "rb").read()
                   for n, _, _, _ in SHIPPED_X4)
        res["outputs_byte_identical_to_serial"] = bool(same)
        # the same loop's forward on the CPU oracle (fp32, as BASELINE configs[0] runs the reference): three of the ten sizes
        # timed, scaled to the ten images by pixel count -- the forward alone (the CPU loop's I/O is the same host code)
        from oracle import codon_oracle as orc
        threads = int(os.environ.get("CODON_CPU_THREADS", str(min(os.cpu_count() or 1, 16))))
        torch.set_num_threads(threads)
        sd = orc.he_state("x4", seed=0)
        sample = [(370, 463), (375, 450), (247, 343)]
        t_cpu, px = 0.0, 0
        with torch.no_grad():
            for h, w in sample:
                x1, y1 = torch.rand((1, 1, h, w)), torch.rand((1, 1, h, w))
                t0 = time.perf_counter()
                orc.forward(sd, x1, y1)
                t_cpu += time.perf_counter() - t0
                px += h * w
        total_px = sum(w * h for _, w, h, _ in SHIPPED_X4)
        res["cpu_oracle_loop_seconds_est"] = t_cpu * total_px / px
        res["cpu_oracle"] = {"kind": "port", "cores": threads, "sample": "one fp32 forward each at 370x463, 375x450, 247x343 "
                             f"({t_cpu:.1f} s), scaled by pixel count to the ten images"}
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def train_leg(model, x, y, dev, dist, rank, world, barrier, steps, warmup, dtype, scale, scaling="weak", ctrl=None,
              data=None):
    """One step = zero_grad, forward, L1 + (1 - SSIM) loss (HIP kernels, forward and backward), backward (HIP
    dgrad/wgrad/CAC kernels), ONE all-reduce of the flat gradient buffer (RCCL when world > 1), Adam step.
    Nothing is skipped.  Returns the result dict on rank 0 (None elsewhere)."""
    from codon_amd.dist import GradSync
    from codon_amd.metrics import L1SSIMLoss
    B, _, H, W = x.shape
    model.train()
    gs = GradSync(model, process_group=data)     # the gradient all-reduce: RCCL (default group) unless the probe failed
    gs.broadcast_parameters(0)
    from codon_amd.dist import FlatAdam
    opt = FlatAdam(gs, lr=1e-4)                  # torch.optim.Adam's update as ONE launch over the flat gradient buffer
    g = torch.Generator(device=dev); g.manual_seed(99 + rank)
    tgt = torch.rand((B, 1, H, W), generator=g, device=dev)
    crit = L1SSIMLoss(1.0, 1.0)          # BASELINE.json configs[2]: L1 + SSIM

    def step():
        gs.zero_grad()
        out = model(x, y)
        loss = crit(out.float(), tgt)
        gs.backward(loss)                # the direct route: the backward's kernels add into the flat all-reduce buffer
        gs.all_reduce_grads()
        opt.step()
        return loss

    from codon_amd import ops
    torch.cuda.reset_peak_memory_stats(dev)
    for _ in range(warmup):
        loss = step()
    # live roofline of the dominant backward kernel: HIP events around every conv5x5 128->128 weight-gradient launch
    ops.PROFILE = {"key": None, "events": [], "wgrad_key": (5, 128, 128), "wgrad_events": []}
    evs = []
    hw = HwmonSampler(dev)
    barrier()
    hw.start()
    t0 = time.perf_counter()
    for _ in range(steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(dev))
        loss = step()
        e1.record(torch.cuda.current_stream(dev))
        evs.append((e0, e1))
    barrier()
    dt = time.perf_counter() - t0
    hw_train = hw.stop(single_rank=(world == 1))
    prof, ops.PROFILE = ops.PROFILE, None
    assert torch.isfinite(loss)
    # one all-reduce of the flat gradient, timed on its own (latency-bound: 7.46 MB over xGMI)
    ar_us = None
    if dist is not None:
        for _ in range(3):
            gs.all_reduce_grads()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(10):
            gs.all_reduce_grads()
        torch.cuda.synchronize(dev)
        ar_us = (time.perf_counter() - t1) / 10 * 1e6
    dt = max_over_ranks(dt, dist, ctrl)
    per_rank = gather_rank_info(step_stats(evs), dist, rank, world, ctrl)
    hw_ranks = gather_rank_info(hw_train, dist, rank, world, ctrl)
    model.check_packed(synchronize=False)
    if rank != 0:
        return None
    P = B * H * W
    step_s = dt / steps
    peak = PEAK_BF16_MFMA_TFLOPS if dtype == "bf16" else PEAK_F32_MFMA_TFLOPS
    tf = 3 * FLOP_PER_PIXEL_FWD * P / step_s / 1e12
    wev = prof["wgrad_events"]
    wms = sum(a.elapsed_time(b) for a, b in wev) / max(len(wev), 1)
    wflop = 2.0 * CONV5_128_MAC_PER_PIXEL * P
    wach = wflop / (wms * 1e-3) / 1e12 if wms > 0 else 0.0
    esz = 2 if dtype == "bf16" else 4
    return {"metric": "iters/sec (fwd+bwd)", "value": steps / dt, "unit": "it/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"CODON x{scale} forward+backward (L1 + SSIM loss, Adam), batch {B}/GPU at {H}x{W}, "
                                   f"{dtype}" + (" activations/gradients, fp32 accumulate + master weights "
                                                 "(BASELINE.json configs[2] per-GPU shape)" if dtype == "bf16" else "") +
                                   "; optimizer = codon_amd.dist.FlatAdam (torch.optim.Adam's update of the 44 fp32 tensors as one "
                                   "codon_adam_step launch over the flat gradient buffer): every launch of the step is a codon_* "
                                   "HIP kernel but the fill that zeroes the gradient buffer",
                       "batch_per_gpu": B, "height": H, "width": W, "global_batch": B * world,
                       "parallelism": f"dp{world}: images sharded, one all-reduce of the flat 1 865 506-element gradient per step"},
            "images_per_s": world * B * steps / dt,
            "step_events": step_stats(evs),
            "per_rank_step_ms": [{k: st_[k] for k in ("median_ms", "min_ms", "max_ms")} if st_ else None for st_ in per_rank],
            "per_rank_hwmon": hw_ranks,
            "whole_step": {"tflops": tf, "frac_mfma_peak": tf / peak,
                           "flop_model": "3 x forward FLOPs (SURVEY.md 8d: dgrad + wgrad per conv)"},
            "roofline": {"bound": "mfma", "kernel": ("conv_wgrad_c8_kernel<5>" if dtype == "bf16" else "conv_wgrad_f32_t16_kernel<5>") +
                                   " 128->128 + its fixed-order reduce (dW of conv3 / conv6 / conv10)",
                         "achieved": wach, "peak": peak, "unit": "TFLOP/s", "frac": wach / peak,
                         "traffic": pmc_traffic("conv_wgrad_c8_kernel<C8Bf16, 5, false, 128, 128", B, H, W,
                                                "r*_bf16_train_b32_480x640_pmc.json") if dtype == "bf16" else None,
                         "traffic_note": "PMC bytes per launch of the 128->128 launches only (the kernel name carries the shape)",
                         "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/)",
                         "traffic_from_hash": pmc_traffic_hash("r*_bf16_train_b32_480x640_pmc.json")[0] if dtype == "bf16" else None,
                         "lib_source_hash": pmc_traffic_hash()[1],
                         "alg_bytes_per_launch": 2 * 128 * esz * P, "launches_timed": len(wev), "avg_launch_ms": wms,
                         "flop_per_launch": wflop},
            "allreduce_us": ar_us, "allreduce_bytes": gs.numel * 4,
            "rccl_ranks": (dist.get_world_size() if dist is not None else 1),
            "loss": float(loss.detach()),
            "peak_mem_gb": torch.cuda.max_memory_allocated(dev) / 1e9}


def _optional(what, fn):
    """The side legs of the default line (shard timings, one-image latencies, the script loop, the CPU leg) must never cost the
    headline: an exception in one of them becomes {"error": ...} in its field and a note on stderr."""
    try:
        return fn()
    except Exception as e:          # noqa: BLE001 -- reported in the line
        print(f"bench.py: optional leg `{what}` failed: {type(e).__name__}: {e}", file=sys.stderr)
        return {"error": f"{type(e).__name__}: {e}"[:500]}


def _timed_ms(fn, dev, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


def strong_shards_fwd(model, x, y, dev, full_ms):
    """one GPU's share of a STRONG-scaling forward (32 images in total over N = 2 / 4 / 8 GPUs).  The forward
    shards images and has no collective, so the N-GPU strong-scaling time IS the one-GPU time of a 32 / N
    batch: measurable here, on one GPU.  efficiency(N) = t(b32) / (N t(b32 / N))."""
    B = x.shape[0]
    ms = {f"b{B}": full_ms}
    with torch.no_grad():
        for n in (2, 4, 8):
            b = B // n
            xs, ys = x[:b].contiguous(), y[:b].contiguous()
            ms[f"b{b}"] = _timed_ms(lambda: model(xs, ys), dev, 3, 1)
    return {"ms_per_step": ms,
            "implied_efficiency": {f"n{n}": full_ms / (n * ms[f"b{B // n}"]) for n in (2, 4, 8)},
            "what": f"fp32 forward of {B} / N images on ONE GPU = a rank's step of --scaling strong at N GPUs (no collective)"}


def strong_shards_train(model, x, y, dev, full_ms):
    """The same for the bf16 training step: forward + L1 + SSIM + backward + Adam on 32 / N images (the flat-gradient all-reduce
    is the only thing an N-GPU step adds: `step_minus_allreduce_budget_ms` says how long it may take)."""
    from codon_amd.dist import FlatAdam, GradSync
    from codon_amd.metrics import L1SSIMLoss
    B = x.shape[0]
    ms = {f"b{B}": full_ms}
    model.train()
    gs = GradSync(model)
    crit = L1SSIMLoss(1.0, 1.0)
    for n in (2, 4, 8):
        b = B // n
        xs, ys = x[:b].contiguous(), y[:b].contiguous()
        g = torch.Generator(device=dev); g.manual_seed(7 + n)
        tgt = torch.rand(xs.shape, generator=g, device=dev)
        opt = FlatAdam(gs, lr=1e-4)

        def step():
            gs.zero_grad()
            gs.backward(crit(model(xs, ys).float(), tgt))
            gs.all_reduce_grads()
            opt.step()

        ms[f"b{b}"] = _timed_ms(step, dev, 5, 2)
        del opt
    eff = {f"n{n}": full_ms / (n * ms[f"b{B // n}"]) for n in (2, 4, 8)}
    return {"ms_per_step": ms, "implied_efficiency_before_allreduce": eff,
            "what": f"bf16 training step on {B} / N images on ONE GPU = a rank's compute of --scaling strong at N GPUs"}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


_RENDEZVOUS_ERRORS = ("EADDRINUSE", "Address already in use", "address already in use", "failed to bind",
                      "The server socket has failed to listen", "Connection refused", "DistNetworkError")


def self_launch(n, argv, timeout_s=None, attempts=3):
    """--gpus N without a launcher: start N fresh rank processes (never an exec of a process that touched the GPU;
    this parent has made no HIP call).  Rank 0 prints the JSON line on our stdout; any failing child fails the run.
    The rendezvous port is picked by bind-and-close, which can race on a busy node: when a child dies of a bind /
    connect error the whole set is terminated (exact PIDs) and started again on another port, up to `attempts` times.
    A wall-clock limit terminates the children and exits non-zero instead of hanging the driver."""
    import subprocess
    import tempfile
    timeout_s = timeout_s or float(os.environ.get("CODON_BENCH_TIMEOUT_S", "1500"))
    rc = 1
    for attempt in range(attempts):
        port = _free_port()
        tmp = tempfile.mkdtemp(prefix="codon_bench_")
        procs, errs = [], []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            ef = open(os.path.join(tmp, f"rank{r}.err"), "w+")
            errs.append(ef)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stderr=ef))
        rc, t0, timed_out = 0, time.time(), False
        live = list(procs)
        while live:
            time.sleep(0.2)
            if time.time() - t0 > timeout_s:
                timed_out = True
                rc = 124
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
            if rc != 0:
                for q in live:              # exact PIDs we started
                    q.terminate()
                deadline = time.time() + 10
                for q in live:
                    try:
                        q.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                live = []
        texts = []
        for r, ef in enumerate(errs):
            ef.seek(0)
            texts.append(ef.read())
            ef.close()
        if rc == 0:
            for t in texts:
                sys.stderr.write(t)
            return 0
        rendezvous = (not timed_out) and any(any(k in t for k in _RENDEZVOUS_ERRORS) for t in texts)
        for r, t in enumerate(texts):
            sys.stderr.write(f"---- rank {r} stderr (attempt {attempt + 1}) ----\n{t[-4000:]}\n")
        if timed_out:
            print(f"bench.py: ranks still running after {timeout_s:.0f} s: terminated", file=sys.stderr)
            return rc
        if not rendezvous or attempt + 1 == attempts:
            print(f"bench.py: a rank exited with code {rc}", file=sys.stderr)
            return rc
        print(f"bench.py: rendezvous on port {port} failed; retrying on another port", file=sys.stderr)
    return rc


def rank_info(dev, local):
    import socket
    props = torch.cuda.get_device_properties(dev)
    return {"rank": int(os.environ.get("RANK", "0")), "local_rank": local, "hostname": socket.gethostname(),
            "device_index": dev.index, "device": props.name, "uuid": str(getattr(props, "uuid", "")),
            "gcn_arch": getattr(props, "gcnArchName", ""), "visible": os.environ.get("HIP_VISIBLE_DEVICES") or
            os.environ.get("ROCR_VISIBLE_DEVICES") or os.environ.get("CUDA_VISIBLE_DEVICES")}


def gather_rank_info(info, dist, rank, world, ctrl=None):
    if dist is None:
        return [info]
    out = [None] * world
    dist.all_gather_object(out, info, group=ctrl)
    return out


def max_over_ranks(v, dist, ctrl=None):
    """The contract's MAX over ranks of a host-side time: a CPU tensor on the control group."""
    if dist is None:
        return float(v)
    t = torch.tensor([v], dtype=torch.float64)          # ctrl is a gloo group (or the default group IS gloo): host tensor
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctrl)
    return float(t.item())


EXIT_RCCL_HUNG = 3      # the line was printed, but an RCCL probe thread hung on some rank: not a healthy run


def rccl_probe(dist, dev, world, ctrl, timeout_s=None):
    """First RCCL collective of the run: one all-reduce of a single element on the default (RCCL) group -- this is where
    the communicator over xGMI is formed.  Never fatal AND bounded: the forward has no data-path collective, so when RCCL
    cannot form a communicator, returns a wrong sum, or HANGS (one rank missing, a stuck fabric) the run continues with every
    collective on the gloo control group and the line says so (`rccl_probe.ok` false, `backend` "gloo").  The collective
    runs on a helper thread that the caller waits for at most CODON_RCCL_PROBE_TIMEOUT_S (default 90 s): a symmetric
    exception, an asymmetric one (the other ranks then wait for the missing peer) and a hang all end at the vote below
    within that time.  All ranks take the same decision (MIN over the control group).  A helper thread still stuck in RCCL
    is left behind (daemon); main() then marks the line `invalid` and leaves through os._exit(EXIT_RCCL_HUNG) so that no
    destructor waits for it and no driver records the run as healthy."""
    import threading
    timeout_s = float(timeout_s if timeout_s is not None else os.environ.get("CODON_RCCL_PROBE_TIMEOUT_S", "90"))
    t0 = time.perf_counter()
    res = {"ok": False, "err": None}

    def work():
        try:
            torch.cuda.set_device(dev)
            t = torch.ones(1, device=dev)
            dist.all_reduce(t)
            torch.cuda.synchronize(dev)
            got = float(t.item())
            res["ok"] = abs(got - world) < 1e-6
            if not res["ok"]:
                res["err"] = f"all_reduce(1) over {world} ranks returned {got}"
        except Exception as e:          # noqa: BLE001 -- reported in the line
            res["err"] = f"{type(e).__name__}: {e}"[:600]

    th = threading.Thread(target=work, name="rccl_probe", daemon=True)
    th.start()
    th.join(timeout_s)
    hung = th.is_alive()
    ok, err = (False, f"no answer from RCCL within {timeout_s:.0f} s (communicator setup or the first all-reduce hangs)") if hung \
        else (res["ok"], res["err"])
    flag = torch.tensor([1.0 if ok else 0.0, 0.0 if hung else 1.0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=ctrl)
    return {"ok": bool(flag[0].item() == 1.0), "this_rank_ok": ok, "error": err, "hung": hung,
            "any_rank_hung": bool(flag[1].item() == 0.0), "ms": (time.perf_counter() - t0) * 1e3, "timeout_s": timeout_s}


def software_versions(backend):
    v = {"torch": torch.__version__, "hip": torch.version.hip}
    try:
        v["rccl"] = ".".join(str(i) for i in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception as e:      # noqa: BLE001 -- version probing must never fail the bench
        v["rccl"] = f"unavailable ({type(e).__name__})"
    return v


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU")
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--scale", type=int, default=4, choices=[4, 8, 16])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fwd-bwd", action="store_true",
                    help="skip the bf16 forward+backward leg (iters/s half of the metric) of the default forward line")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default). gloo only to rehearse the multi-rank path on one GPU")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: BASELINE configs[1] (default); bf16: bf16 activations/weights, fp32 accumulate + master weights")
    ap.add_argument("--conv-precision", choices=["exact", "f16x3"], default="exact",
                    help="fp32 only. exact (default): fp32 MFMA. f16x3: OPT-IN split-precision convs (3 f16 MFMAs per "
                         "product, fp32 accumulate, ~2^-22 per product) -- reported with its own dtype label")
    ap.add_argument("--model", choices=["codon", "rmcr"], default="codon",
                    help="codon (default): CODONNet. rmcr: the conv-only ablation BaseNet_RMCR_fuseRMCR "
                         "(CODON_X16/CODON_x16.py:16-90, SURVEY 8f row f4) -- same 19 convs, no CAC gates: the pure-conv "
                         "roofline probe; forward only")
    ap.add_argument("--mode", choices=["fwd", "train"], default="fwd",
                    help="fwd: BASELINE metric (maps/s); train: fwd + L1+SSIM loss + bwd + grad all-reduce + Adam step (iters/s)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): --batch images PER GPU whatever N. strong: --batch images IN TOTAL, split over the "
                         "N ranks (SURVEY.md 8e); the label goes into the JSON line")
    ap.add_argument("--rccl-selfcheck", choices=["auto", "on", "off"], default="auto",
                    help="before timing, run SURVEY 8(e)'s gradient-equality check on the product kernels over the process "
                         "group (codon_amd.dist.grad_equality_selfcheck): `grad_equal` in rank 0's line.  auto = when N > 1")
    ap.add_argument("--no-strong-shards", action="store_true",
                    help="skip the strong-scaling shard timings (b16 / b8 / b4 on one GPU) of the N = 1 default line")
    ap.add_argument("--no-script-pattern", action="store_true",
                    help="skip the single-image latencies at the reference script's image sizes (N = 1 default line only)")
    a = ap.parse_args()
    if os.environ.get("CODON_BENCH_DUMP_S"):
        # diagnostics: every N seconds every thread's Python stack goes to stderr (where is a slow or hung rank?)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CODON_BENCH_DUMP_S"]), repeat=True, file=sys.stderr)
    if a.scaling == "strong":
        if a.batch % a.gpus != 0:
            raise SystemExit(f"bench.py: --scaling strong needs --batch ({a.batch}) divisible by --gpus ({a.gpus})")

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        raise SystemExit(self_launch(a.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    ndev = torch.cuda.device_count()
    share = a.backend == "gloo" or os.environ.get("CODON_BENCH_SHARE_GPU") == "1"   # rehearsal: ranks may share a GPU
    local = local % max(ndev, 1) if share else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = ctrl = data = probe = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost") and os.path.isdir("/sys/class/net/lo"):
            # one node by contract: keep gloo off the container hostname (it may not resolve) and on the loopback device
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        if a.backend == "nccl":
            # Default group = RCCL (its communicator forms at the first GPU collective: rccl_probe).  The CONTROL plane --
            # the barriers around the timed region, the max over ranks, the rank records -- runs on a gloo group beside
            # it: the forward shards images and has no data-path collective, so its measurement must not depend on the
            # health of a fabric it does not use.  Training's gradient all-reduce and the self-check go through RCCL.
            # a collective that times out raises in the waiting thread instead of tearing the process down (the probe's
            # fallback needs the process alive); the bench still ends non-zero if a gradient all-reduce ever times out
            os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "2")
            dist.init_process_group("nccl", timeout=datetime.timedelta(seconds=int(os.environ.get("CODON_RCCL_TIMEOUT_S", "600"))))
            ctrl = dist.new_group(backend="gloo")
            probe = rccl_probe(dist, dev, world, ctrl)
            data = None if probe["ok"] else ctrl
            if not probe["ok"] and rank == 0:
                print(f"bench.py: RCCL probe failed ({probe['error']}); collectives fall back to gloo", file=sys.stderr)
        else:
            dist.init_process_group("gloo")

    from codon_amd import BaseNet_RMCR_fuseRMCR, CODONNet, CODONNet16, ops
    B, H, W = (a.batch // world if a.scaling == "strong" else a.batch), a.height, a.width
    selfcheck = None
    if a.rccl_selfcheck == "on" or (a.rccl_selfcheck == "auto" and world > 1):
        # first contact with the multi-rank path: is the averaged gradient the single-process gradient?  (never fatal:
        # the line is still printed, with grad_equal false and the reason)
        from codon_amd.dist import grad_equality_selfcheck
        try:
            selfcheck = grad_equality_selfcheck(dev, group=data)
        except Exception as e:          # noqa: BLE001
            selfcheck = {"grad_equal": False, "error": f"{type(e).__name__}: {e}"[:500]}
        torch.cuda.empty_cache()
    ranks = gather_rank_info(rank_info(dev, local), dist, rank, world, ctrl)
    versions = software_versions(a.backend if world > 1 else None)
    torch.manual_seed(0)
    rmcr = a.model == "rmcr"
    if rmcr and a.mode != "fwd":
        raise SystemExit("bench.py: --model rmcr is forward only")
    model = (BaseNet_RMCR_fuseRMCR if rmcr else CODONNet16 if a.scale == 16 else CODONNet)().to(dev).eval()   # reference init rule, seed 0
    bf16 = a.dtype == "bf16"
    if bf16:
        model.set_compute_dtype(torch.bfloat16)
    split = a.conv_precision == "f16x3" and not bf16
    if split:
        model.set_conv_precision("f16x3")
    esize = 2 if bf16 else 4
    peak_mfma = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    x, y = synth_inputs(B, H, W, a.scale, 1234 + rank, dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier(group=ctrl)
        torch.cuda.synchronize(dev)

    if a.mode == "train":
        res = train_leg(model, x, y, dev, dist, rank, world, barrier, a.steps, a.warmup, a.dtype, a.scale, a.scaling,
                        ctrl=ctrl, data=data)
        if rank == 0:
            for r_, st_, hw_ in zip(ranks, res["per_rank_step_ms"], res["per_rank_hwmon"]):
                r_["train_step_ms"] = st_
                r_["train_power_w_p50"], r_["train_sclk_mhz_p50"] = hw_["power_w_p50"], hw_["sclk_mhz_p50"]
                r_["power_cap_w"] = hw_["power_cap_w"]
            res["ranks"], res["versions"] = ranks, versions
            res["backend"] = None if dist is None else ("gloo" if (a.backend == "gloo" or data is not None) else "nccl")
            res["control_backend"] = None if dist is None else "gloo"
            if probe is not None:
                res["rccl_probe"] = probe
            if selfcheck is not None:
                res["grad_equal"], res["rccl_selfcheck"] = selfcheck["grad_equal"], selfcheck
            print(json.dumps(res), flush=True)
        if dist is not None:
            dist.barrier(group=ctrl)
            dist.destroy_process_group()
        return

    with torch.no_grad():
        for _ in range(a.warmup):
            out = model(x, y)
        model.check_packed()                # natural sync point: no forward so far ran on stale packed weights
        ops.PROFILE = {"key": (5, 128, 128), "events": []}
        dtype_label = "f32 via 3xf16-split MFMA (opt-in, not exact fp32)" if split else a.dtype
        step_ev = []
        hw = HwmonSampler(dev)
        barrier()
        hw.start()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(dev))
            out = model(x, y)
            e1.record(torch.cuda.current_stream(dev))
            step_ev.append((e0, e1))
        barrier()
        dt = time.perf_counter() - t0
        hw_fwd = hw.stop(single_rank=(world == 1))
        prof, ops.PROFILE = ops.PROFILE, None
    assert torch.isfinite(out).all()

    dt = max_over_ranks(dt, dist, ctrl)
    fwd_per_rank = gather_rank_info(step_stats(step_ev), dist, rank, world, ctrl)   # a straggler GPU shows up per rank
    hw_per_rank = gather_rank_info(hw_fwd, dist, rank, world, ctrl)                 # ... and a capped socket as a lower clock

    res = None
    if rank == 0:
        P = B * H * W
        maps_s = world * B * a.steps / dt
        ev = prof["events"]
        kms = sum(s.elapsed_time(e) for s, e in ev) / max(len(ev), 1)
        chained = bool(prof.get("chained"))       # the launch also carries the 128->64 1x1 (+2 % MACs)
        kflop = 2.0 * (CONV5_128_MAC_PER_PIXEL + (128 * 64 if chained else 0)) * P
        ach = kflop / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        step_s = dt / a.steps
        res = {
            "metric": "HR depth maps/sec (fwd)", "value": maps_s, "unit": "maps/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": step_s * 1e3, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": dtype_label, "data": "synthetic",
            "config": {"workload": f"CODON x{a.scale} forward, batch {B}/GPU at {H}x{W}, fp32 "
                                   f"(BASELINE.json configs[1])" if (B, H, W, a.scale, bf16, split) == (32, 480, 640, 4, False, False)
                       else f"CODON x{a.scale} forward, batch {B}/GPU at {H}x{W}, {dtype_label}",
                       "batch_per_gpu": B, "global_batch": B * world, "height": H, "width": W,
                       "parallelism": f"dp{world}: images sharded across ranks, no collective in forward",
                       "weights": "reference init rule (He-normal convs, default CAC), torch.manual_seed(0)"},
            "step_events": step_stats(step_ev),
            "roofline": {"bound": "mfma", "kernel": ("conv_mfma_f32x3_kernel<5,128>" if split else
                                                      "conv_c8_kernel<bf16,5,128,128> (channel-blocked activations)" if bf16 else
                                                      "conv_mfma_f32_kernel<5,128,128>") +
                                   (" + register-chained 1x1 128->64 (conv3+confuse / conv6+confuse_c / conv10+confuse_fuse)"
                                    if chained else " (conv3/conv6/conv10)"),
                         "achieved": ach * (3 if split else 1), "peak": PEAK_BF16_MFMA_TFLOPS if split else peak_mfma,
                         "unit": "TFLOP/s", "frac": ach * (3 if split else 1) / (PEAK_BF16_MFMA_TFLOPS if split else peak_mfma),
                         "note": "f16x3: achieved counts the 3 f16 MFMA products actually issued per fp32 product" if split else None,
                         "traffic": (pmc_traffic("conv_c8_kernel<C8Bf16, 5, 128, 128, true", B, H, W,
                                                 "r*_bf16_fwd_b32_480x640_pmc.json") if bf16 else
                                     None if split else pmc_traffic("codon::conv_mfma_f32_kernel<5, 128, 128", B, H, W)),
                         "traffic_unit": "bytes/launch (rocprofv3 PMC, profiles/)",
                         "traffic_from_hash": None if split else pmc_traffic_hash(
                             pmc_pattern(B, H, W, "r*_bf16_fwd_b32_480x640_pmc.json" if bf16 else "r*_fwd_b32_480x640_pmc*.json")
                             or "none")[0],
                         "lib_source_hash": pmc_traffic_hash()[1],
                         "alg_bytes_per_launch": (128 + 64 if chained else 2 * 128) * esize * P,
                         "launches_timed": len(ev), "avg_launch_ms": kms,
                         "flop_per_launch": kflop},
            "whole_forward": {"tflops": FLOP_PER_PIXEL_FWD * P / step_s / 1e12,
                              "frac_mfma_peak": FLOP_PER_PIXEL_FWD * P / step_s / 1e12 / peak_mfma,
                              "alg_hbm_gbs": ALG_ELEMS_PER_PIXEL * esize * P / step_s / 1e9,
                              "frac_hbm_peak": ALG_ELEMS_PER_PIXEL * esize * P / step_s / 1e9 / PEAK_HBM_GBS,
                              "mpx_per_s": world * P / step_s / 1e6},
        }
        default_line = world == 1 and (B, H, W, a.scale) == (32, 480, 640, 4) and not bf16 and not split and not rmcr
        if default_line and not a.no_strong_shards:
            res["strong_shards"] = {"fwd_fp32": _optional("strong_shards.fwd_fp32", lambda: strong_shards_fwd(model, x, y, dev, step_s * 1e3))}
        if world == 1 and not bf16 and not split and a.mode == "fwd" and not rmcr:
            # OPT-IN mode, reported beside (never instead of) the exact-fp32 headline: same inputs, same K steps
            model.set_conv_precision("f16x3")
            with torch.no_grad():
                o3 = model(x, y)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(a.steps):
                    o3 = model(x, y)
                torch.cuda.synchronize(dev)
                dt3 = (time.perf_counter() - t1) / a.steps
            model.set_conv_precision("exact")
            dev_rmse = float((o3.double() - out.double()).pow(2).mean().sqrt())
            res["optin_f16x3"] = {
                "what": "model.set_conv_precision('f16x3'): 3x3/5x5 convs as 3 f16 MFMAs per product on fp16 hi+lo "
                        "splits of the fp32 operands, fp32 accumulate; NOT the headline value above",
                "value": B / dt3, "unit": "maps/s", "ms_per_step": dt3 * 1e3,
                "rmse_vs_exact_fp32_output": dev_rmse, "output_std": float(out.double().std()),
                "parity": "passes the same RMSE <= 1e-4 fixtures as the exact path (tests/test_gpu_f16x3.py)"}
        res["rccl_ranks"] = dist.get_world_size() if dist is not None else 1
        # `backend`: what the data-path collectives (training's gradient all-reduce, the self-check) run on; the control
        # plane (barriers, max over ranks, rank records) is gloo whenever RCCL is the data backend
        res["backend"] = None if dist is None else ("gloo" if (a.backend == "gloo" or data is not None) else "nccl")
        res["control_backend"] = None if dist is None else "gloo"
        if probe is not None:
            res["rccl_probe"] = probe
        for r_, st_, hw_ in zip(ranks, fwd_per_rank, hw_per_rank):
            r_["fwd_step_ms"] = {k: st_[k] for k in ("median_ms", "min_ms", "max_ms")} if st_ else None
            r_.update(hw_)           # power_w_p50 / power_w_p95 / sclk_mhz_p50 / power_cap_w of the forward's timed region
        res["ranks"], res["versions"] = ranks, versions
        if selfcheck is not None:
            res["grad_equal"], res["rccl_selfcheck"] = selfcheck["grad_equal"], selfcheck
    if rank == 0 and rmcr:
        # no CAC gates: 5 x (518 activation elements + 250 MAC of the 5x5 2->1 spatial conv) less per pixel
        step_s_ = dt / a.steps
        fl_, el_ = FLOP_PER_PIXEL_FWD - 5 * 500, ALG_ELEMS_PER_PIXEL - 5 * 518
        res["whole_forward"].update({"tflops": fl_ * B * H * W / step_s_ / 1e12,
                                     "frac_mfma_peak": fl_ * B * H * W / step_s_ / 1e12 / peak_mfma,
                                     "alg_hbm_gbs": el_ * esize * B * H * W / step_s_ / 1e9,
                                     "frac_hbm_peak": el_ * esize * B * H * W / step_s_ / 1e9 / PEAK_HBM_GBS})
        res["config"]["workload"] = res["config"]["workload"].replace("CODON x", "BaseNet_RMCR_fuseRMCR (conv-only ablation, no CAC gates) x")
        res["metric"] = "HR depth maps/sec (fwd), conv-only ablation"
    if not bf16 and not split and not a.no_fwd_bwd and not rmcr:
        # second half of the BASELINE metric, every rank takes part (the step holds the RCCL all-reduce): configs[2]'s
        # per-GPU shape -- same net, same batch/GPU and image size, bf16 activations, fp32 accumulate + master weights
        del out
        model = None
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        tm = (CODONNet16 if a.scale == 16 else CODONNet)().to(dev)
        tm.set_compute_dtype(torch.bfloat16)
        tsteps = 10 if (B, H, W) == (32, 480, 640) else max(3, min(a.steps, 10))     # >= 10 timed iterations
        try:
            leg = train_leg(tm, x, y, dev, dist, rank, world, barrier, tsteps, 2, "bf16", a.scale, a.scaling, ctrl=ctrl, data=data)
        except Exception as e:          # noqa: BLE001
            if world > 1:               # the other ranks are inside the step's collectives: nothing to salvage
                raise
            # one rank: the forward line (the headline `value`) is still printed, with the reason the second half is missing
            print(f"bench.py: the fwd+bwd leg failed: {type(e).__name__}: {e}", file=sys.stderr)
            res["fwd_bwd"] = {"error": f"{type(e).__name__}: {e}"[:500]}
            leg = None
        shards_train = None
        if leg is not None and rank == 0 and res is not None and "strong_shards" in res:
            torch.cuda.empty_cache()
            shards_train = _optional("strong_shards.train_bf16", lambda: strong_shards_train(tm, x, y, dev, leg["ms_per_step"]))
        del tm
        torch.cuda.empty_cache()
        if rank == 0 and leg is not None:
            res["fwd_bwd"] = {"it_per_s": leg["value"], "unit": "it/s (whole job: one optimizer step over the global "
                              "batch per iteration)", "images_per_s": leg["images_per_s"], "ms_per_step": leg["ms_per_step"],
                              "steps": leg["steps"], "warmup": leg["warmup"], "dtype": "bf16",
                              "workload": leg["config"]["workload"], "global_batch": leg["config"]["global_batch"],
                              "tflops": leg["whole_step"]["tflops"], "frac": leg["whole_step"]["frac_mfma_peak"],
                              "peak": PEAK_BF16_MFMA_TFLOPS, "rccl_ranks": leg["rccl_ranks"], "loss": leg["loss"],
                              "peak_mem_gb": leg["peak_mem_gb"], "step_events": leg["step_events"],
                              "roofline": leg["roofline"], "allreduce_us": leg["allreduce_us"],
                              "allreduce_bytes": leg["allreduce_bytes"], "scaling": leg["scaling"]}
            if world == 1:
                t1_ = leg["ms_per_step"]
                budget = {"weak_8gpu_ge_6x": t1_ * (8.0 / 6.0 - 1.0),
                          "what": "north_star: >= 6x at 8 GPUs.  Weak scaling (32 images per GPU, the driver's line): the 8-GPU "
                                  "step may take 8/6 of this one-GPU step, i.e. the RCCL all-reduce of the 7.46 MB gradient "
                                  "(fwd_bwd.allreduce_us in an N > 1 line) plus any straggler may add this many ms"}
                if shards_train is not None:
                    res["strong_shards"]["train_bf16"] = shards_train
                if shards_train is not None and "ms_per_step" in shards_train:
                    budget["strong_8gpu_ge_6x"] = t1_ / 6.0 - shards_train["ms_per_step"]["b4"]
                    budget["what"] += "; strong scaling (32 images in total): step(b4) + all-reduce must stay below step(b32) / 6"
                res["fwd_bwd"]["step_minus_allreduce_budget_ms"] = budget
            for r_, st_, hw_ in zip(res["ranks"], leg["per_rank_step_ms"], leg["per_rank_hwmon"]):
                r_["train_step_ms"] = st_
                r_["train_power_w_p50"], r_["train_sclk_mhz_p50"] = hw_["power_w_p50"], hw_["sclk_mhz_p50"]
    if rank == 0:
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = _optional("cpu_baseline", lambda: cpu_baseline(H, W))
            if not bf16 and not split and not rmcr:
                res["config0_on_gpu"] = _optional("config0_on_gpu", lambda: config0_gpu_latency(dev))
                if not a.no_script_pattern:
                    res["script_pattern_on_gpu"] = _optional("script_pattern_on_gpu", lambda: script_pattern_latency(dev))
                    loop = _optional("script_loop", lambda: script_loop_throughput(dev))
                    res["script_pattern_on_gpu"]["script_loop"] = loop
                    res["script_pattern_on_gpu"]["script_loop_images_per_s"] = loop.get("script_loop_images_per_s")
        if probe is not None and probe.get("any_rank_hung"):
            # a helper thread was still stuck inside an RCCL collective on some rank while the legs were timed --
            # it may have held CUs.  The line says so and the process exits with EXIT_RCCL_HUNG: a driver that looks at the
            # exit code alone must not record a healthy run on a node whose fabric hangs.
            res["invalid"] = ("an RCCL probe thread was stuck in a collective on at least one rank during the timed legs "
                              f"(rccl_probe.any_rank_hung): timings may be contaminated; exit code {EXIT_RCCL_HUNG}")
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier(group=ctrl)
        if probe is not None and probe.get("any_rank_hung"):
            # a probe thread is still inside RCCL on some rank: no destructor (here or on a peer) may wait for it
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(EXIT_RCCL_HUNG)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
