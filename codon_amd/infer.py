"""python -m codon_amd.infer -- the reference's test loop (CODON_X4/test.py:60-145) on MI355X:
for every image: read depth (already HR-sized) + guidance, forward, clip/*255/uint8, write PNG, masked RMSE
vs the label, SSIM vs the label; print per-image values and the means.  Everything numeric runs in HIP kernels
(codon_amd.CODONNet, codon_amd.metrics); this file is I/O glue.

Round 6: the loop is a PIPELINE.  One forward of a Middlebury image takes 2.3 ms; decoding its two PNGs, the float
conversion, the upload, the download and the PNG encode take several times that on the host, and the reference's loop
(test.py:109-145) does all of it serially per image.  Here
  reader thread : decode image i+1 (PIL releases the GIL), /255, cast, pinned host tensors, upload on a SIDE stream
  main thread   : forward + post-processing + metrics of image i on the caller's stream (waits for the upload's event)
  writer thread : download of image i-1 has landed in pinned memory (event) -> PNG encode + write
Same arithmetic per image in the same order: outputs are byte-identical to the serial loop (`--serial`, kept for the A/B
and the test)."""
from __future__ import annotations

import argparse
import math
import os
import queue
import threading
import time

import numpy as np
import torch

from . import CODONNet, CODONNet16, io, metrics


def list_pairs(input_depth: str, input_color: str):
    return [f for f in sorted(os.listdir(input_color)) if os.path.exists(os.path.join(input_depth, f))]


def _load_host(a_depth, a_color, a_label, f, tdt):
    """Host side of one image: decode, grey, /255 (float64 divide, then float32: test.py:116-123), crop both to the common
    size, cast to the model's dtype (the rounding `.cuda().half()` does on the device, done here on half the bytes)."""
    # numpy only (single-threaded): torch's CPU ops fan a 170 k-element conversion out over every host core, which costs
    # milliseconds per call on a 128-core box; the values are io.to_input()'s -- float64 divide, float32, then the dtype's
    # round-to-nearest-even -- bit for bit
    px = io.read_gray(os.path.join(a_depth, f))
    py = io.read_gray(os.path.join(a_color, f))
    h, w = min(px.shape[0], py.shape[0]), min(px.shape[1], py.shape[1])
    ndt = {torch.float32: np.float32, torch.float16: np.float16}.get(tdt)

    def conv(p_):
        v = (np.asarray(p_[:h, :w]) / 255).astype(np.float32)
        if ndt is not None:
            return torch.from_numpy(np.ascontiguousarray(v.astype(ndt)))[None, None]
        # bf16 has no numpy type: round to nearest even on the bit pattern (finite, non-negative inputs)
        u = np.ascontiguousarray(v).view(np.uint32)
        b16 = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
        return torch.from_numpy(b16.view(np.int16)).view(torch.bfloat16)[None, None]

    lab = torch.from_numpy(io.read_gray(os.path.join(a_label, f)).copy()) if a_label else None
    return conv(px), conv(py), lab, h, w


READERS = int(os.environ.get("CODON_INFER_READERS", "3"))


def _pinned_copy(t: torch.Tensor) -> torch.Tensor:
    """t in page-locked memory, filled by a numpy copy: Tensor.pin_memory() copies with an at::parallel_for, i.e. spins up an
    OpenMP team of every host core in whichever thread calls it first (measured on the 128-core GPU box: 130 ms per image in a
    fresh reader thread)."""
    hp = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    if t.dtype == torch.bfloat16:
        hp.view(torch.int16).numpy()[...] = t.view(torch.int16).numpy()
    else:
        hp.numpy()[...] = t.numpy()
    return hp


def run_loop(model, dev, tdt, input_depth, input_color, label=None, out_dir=None, pipelined=True, emit=print, files=None):
    """The test loop over every image pair.  Returns {"n", "rmse_mean", "ssim_mean", "seconds", "images_per_s"}."""
    files = list_pairs(input_depth, input_color) if files is None else files
    t0 = time.perf_counter()
    rm_sum = ss_sum = 0.0
    n = 0

    def finish(f, out, lab, h, w):
        nonlocal rm_sum, ss_sum, n
        out_u8 = metrics.postprocess_u8(out[0, 0])
        line = f
        if lab is not None:
            rm = metrics.masked_rmse(lab, out_u8)
            ss = metrics.ssim(lab[:h, :w].float() / 255, out_u8.float() / 255)
            rm_sum += rm; ss_sum += ss
            line += f" {rm} {ss}"
        n += 1
        return out_u8, line

    if not pipelined:
        for f in files:
            x, y, lab, h, w = _load_host(input_depth, input_color, label, f, tdt)
            with torch.no_grad():
                out = model(x.to(dev), y.to(dev))
            out_u8, line = finish(f, out, lab.to(dev) if lab is not None else None, h, w)
            if out_dir:
                io.write_gray(os.path.join(out_dir, f), out_u8.cpu().numpy())
            emit(line)
    else:
        main_s = torch.cuda.current_stream(dev)
        up_s = torch.cuda.Stream(device=dev)
        # READERS reader threads take every READERS-th image each (decoding two PNGs is the longest stage: 3.1 ms against a
        # 2.3 ms forward); the main thread takes their queues in turn, so images are consumed in order
        q_ins = [queue.Queue(maxsize=2) for _ in range(READERS)]
        q_out: "queue.Queue" = queue.Queue(maxsize=4)
        errs = []
        stop = threading.Event()

        def reader(k):
            q_in = q_ins[k]
            try:
                torch.cuda.set_device(dev)
                for f in files[k::READERS]:
                    if stop.is_set():
                        break
                    x, y, lab, h, w = _load_host(input_depth, input_color, label, f, tdt)
                    host = [_pinned_copy(t) for t in ((x, y) if lab is None else (x, y, lab))]
                    with torch.cuda.stream(up_s):
                        devs = [t.to(dev, non_blocking=True) for t in host]
                        ev = torch.cuda.Event()
                        ev.record(up_s)
                    q_in.put((f, devs, host, ev, h, w))          # `host` rides along: pinned sources stay alive until consumed
            except BaseException as e:      # noqa: BLE001 -- re-raised in the main thread
                errs.append(e)
            finally:
                q_in.put(None)

        def writer():
            try:
                while True:
                    item = q_out.get()
                    if item is None:
                        return
                    f, host_u8, ev = item
                    ev.synchronize()                               # the download of this image has landed
                    if out_dir:
                        io.write_gray(os.path.join(out_dir, f), host_u8.numpy())
            except BaseException as e:      # noqa: BLE001
                errs.append(e)

        def settle(pend):
            """Metrics of an image whose launches were enqueued one image ago: its three numbers have been downloaded behind
            its kernels; the same host arithmetic as metrics.masked_rmse / metrics.ssim."""
            nonlocal rm_sum, ss_sum, n
            f, h_acc, h_ss, ev = pend
            line = f
            if h_acc is not None:
                ev.synchronize()
                s_, c_ = (int(v) for v in h_acc)
                rm, ss = math.sqrt(s_ / c_), float(h_ss.item())
                rm_sum += rm; ss_sum += ss
                line += f" {rm} {ss}"
            n += 1
            emit(line)

        trs = [threading.Thread(target=reader, args=(k,), name=f"codon_infer_reader{k}") for k in range(READERS)]
        tw = threading.Thread(target=writer, name="codon_infer_writer")
        for t_ in trs + [tw]:
            t_.start()
        pending = None
        try:
            for i in range(len(files)):
                item = q_ins[i % READERS].get()
                if item is None:                                   # that reader failed: its error is re-raised below
                    break
                f, devs, _host, ev, h, w = item
                main_s.wait_event(ev)
                for t in devs:
                    t.record_stream(main_s)                        # allocated on the upload stream, consumed on this one
                with torch.no_grad():
                    out = model(devs[0], devs[1])
                out_u8 = metrics.postprocess_u8(out[0, 0])
                h_acc = h_ss = None
                if len(devs) > 2:                                  # metrics stay on the device; read back one image later
                    lab = devs[2]
                    acc = metrics.masked_sqerr_dev(lab, out_u8)
                    ssv = metrics.ssim_dev(lab[:h, :w].float() / 255, out_u8.float() / 255)
                    h_acc = torch.empty(2, dtype=torch.int64, pin_memory=True)
                    h_ss = torch.empty(1, dtype=torch.float64, pin_memory=True)
                    h_acc.copy_(acc, non_blocking=True)
                    h_ss.copy_(ssv, non_blocking=True)
                host_u8 = torch.empty(out_u8.shape, dtype=torch.uint8, pin_memory=True)
                host_u8.copy_(out_u8, non_blocking=True)
                dv = torch.cuda.Event()
                dv.record(main_s)
                q_out.put((f, host_u8, dv))
                if pending is not None:
                    settle(pending)                                # image i-1, while image i runs
                pending = (f, h_acc, h_ss, dv)
            if pending is not None:
                settle(pending)
        finally:
            q_out.put(None)
            stop.set()
            for t_, q_ in zip(trs, q_ins):                         # error paths: a reader may be blocked on a full queue
                while t_.is_alive():
                    try:
                        q_.get_nowait()
                    except queue.Empty:
                        pass
                    t_.join(0.02)
            tw.join()
        if errs:
            raise errs[0]
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"n": n, "rmse_mean": rm_sum / n if (label and n) else None, "ssim_mean": ss_sum / n if (label and n) else None,
            "seconds": dt, "images_per_s": n / dt if dt > 0 else 0.0}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--scale", type=int, default=4, choices=[4, 8, 16])
    ap.add_argument("--input-depth", required=True)
    ap.add_argument("--input-color", required=True)
    ap.add_argument("--label", default=None)
    ap.add_argument("--out", default=None)
    ap.add_argument("--weights", default=None, help="X4.pth-style checkpoint; random reference init if absent")
    ap.add_argument("--dtype", default="f16", choices=["f32", "bf16", "f16"], help="the reference script runs .half()")
    ap.add_argument("--serial", action="store_true", help="the reference's own serial loop (decode, upload, forward, download, "
                                                          "encode one after the other per image) instead of the pipeline")
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit("No GPU found, codon_amd has no CPU path")          # test.py:37-38
    dev = torch.device("cuda:0")
    model = (CODONNet16 if a.scale == 16 else CODONNet)()
    if a.weights:
        print("loaded epoch", io.load_checkpoint(a.weights, model))
    else:
        print("WARNING: no --weights given (the reference's X*.pth are not shipped): random init")
    tdt = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    model = model.to(dev).to(tdt).eval()
    if a.out:
        os.makedirs(a.out, exist_ok=True)
    r = run_loop(model, dev, tdt, a.input_depth, a.input_color, a.label, a.out, pipelined=not a.serial)
    print(r["n"])
    if a.label and r["n"]:
        print(r["rmse_mean"], r["ssim_mean"])
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
