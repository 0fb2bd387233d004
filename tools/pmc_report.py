"""Merge rocprofv3 PMC passes of ONE bench command into a per-kernel table (profiles/rNN_*_pmc.json):

  python tools/pmc_report.py <fetch_dir> <write_dir> <busy_dir|-> <out.json> [B H W [mode [esize]]]

  fetch_dir : rocprofv3 --kernel-trace --pmc FETCH_SIZE ...
  write_dir : rocprofv3 --kernel-trace --pmc WRITE_SIZE ...                       (separate pass: TCC has 4 slots)
  busy_dir  : rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES ...   ("-" to skip)
  mode      : fwd (default) | train -- which roles the kernels play in the profiled command (a training step runs the
              chained conv with its 128-channel `mid` output and the conv kernels as ReLU-masked dgrads)
  esize     : bytes per activation element, 2 (default: bf16 / fp16) or 4 (fp32 NCHW kernels)

Corrections (MI355X_MICROARCH.md, HBM section): both sizes are reported in KB; FETCH_SIZE tallies 128-B requests at
64 B on gfx950, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  clock = GRBM_GUI_ACTIVE / 8 XCDs /
kernel time; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x per-XCD active cycles).

Joining the passes (round 4): the command is deterministic, so the i-th dispatch of a kernel in one pass IS the i-th
dispatch of that kernel in every other pass.  Rows are kept per dispatch (Dispatch_Id order) and joined by that index,
never by name order; a kernel whose launch count differs between passes is reported with "pass_mismatch" and gets no
ratio.  Clock and busy are formed per dispatch from the two counters of THAT dispatch; dispatches whose clock falls
outside 0.5-3.0 GHz or whose busy fraction exceeds 1 are dropped from the average and counted in "busy_pass_rejected" (a
kernel with more than a quarter of its dispatches rejected carries "suspect": true and no clock / busy at all).  The r03
table held a row at 203 GHz: ONE of that kernel's 8 dispatches -- the first kernel after a long idle gap -- reports a
GRBM_GUI_ACTIVE 1000x too large (26.5e9 instead of 26e6), and a plain average kept it.  Algorithmic bytes are attached only where every launch under that kernel name plays the same role in the
profiled command (a like-for-like ratio); a name that mixes roles (forward conv + dgrad + accumulate variants of one
instantiation in a training step) gets "roles": "mixed" and no ratio."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def clean(name):
    return re.sub(r"\(.*$", "", name).replace("void ", "").replace("codon::", "").strip()


def load(d, counters):
    """kernel name -> list (dispatch order) of {counter: value, "ns": duration}."""
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    per = {}                                           # (file, dispatch id) -> record
    for f in sorted(files):
        for row in csv.DictReader(open(f)):
            c = row.get("Counter_Name")
            if c not in counters:
                continue
            key = (f, int(row["Dispatch_Id"]))
            r = per.setdefault(key, {"name": clean(row["Kernel_Name"]),
                                     "ns": float(row["End_Timestamp"]) - float(row["Start_Timestamp"])})
            r[c] = r.get(c, 0.0) + float(row["Counter_Value"])       # a counter may come as one row per instance
    acc = defaultdict(list)
    for key in sorted(per):
        acc[per[key]["name"]].append(per[key])
    return acc


def alg_bytes(name, P, es, mode):
    """(bytes the kernel must move once per launch, role note) or (None, reason)."""
    train = mode == "train"
    m = re.match(r"conv_c8_kernel<C8\w+, (\d), (\d+), (\d+), (true|false)(?:, \d+, (true|false))?(?:, (?:true|false), (?:true|false))?(?:, \d+)?>", name)
    if m:
        k, ci, co, fuse, gate = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4) == "true", m.group(5) == "true"
        if fuse:      # conv5x5 + chained 1x1: 128 in + 64 out; a training step also writes the 128-channel mid
            return ((ci + 64 + (128 if train else 0)) * es * P, "chained 5x5+1x1" + (" + mid" if train else ""))
        if gate:      # gated staging reads pre AND inputs; an emitting launch also writes the gated tensor: in the default
            # schedule (model.GATED_EMIT) every gated conv5x5 emits, conv7 (3x3 128->64) emits in a training step only
            emit = k == 5 or train
            return (((3 if emit else 2) * ci + co) * es * P, "gated staging" + (" + emitted gated input" if emit else ""))
        if train and (k, ci, co) == (5, 128, 128):
            return ((ci + 2 * co) * es * P, "dgrad of conv3/6/10: gy in, ReLU mask in, gx out")
        if train:
            return (None, "mixed")                     # forward conv, dgrad, dgrad + accumulate share the instantiation
        return ((ci + co) * es * P, "forward conv")
    m = re.match(r"conv_mfma_f32_kernel<(\d), (\d+), (\d+), (\d+), (true|false), (true|false)", name)
    if m:
        k, ci, co, fuse, gate = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(5) == "true", m.group(6) == "true"
        if train:
            return (None, "mixed")
        if fuse:
            return ((ci + 64) * 4 * P, "chained 5x5+1x1")
        if gate:
            emit = k == 5
            return (((3 if emit else 2) * ci + co) * 4 * P, "gated staging" + (" + emitted gated input" if emit else ""))
        return ((ci + co) * 4 * P, "forward conv")
    m = re.match(r"conv_wgrad_c8_kernel<C8\w+, (\d)(?:, (true|false))?(?:, (\d+), (\d+))?(?:, (true|false))?>", name)
    if m:
        k, dg, gb = int(m.group(1)), m.group(2) == "true", m.group(5) == "true"
        if k == 1:       # GB: gy is the block's dL/d(out), dL/d(pre) formed while staging: + four fp32 / int32 maps per pixel
            return ((128 + 64 + (128 if dg else 0)) * es * P + (16 * P if gb else 0),
                    "1x1 128->64: x, gy in" + (", masked gx out" if dg else "") + (", gate-backward maps in" if gb else ""))
        if m.group(3) and int(m.group(3)) > 0:
            ci, co = int(m.group(3)), int(m.group(4))
            return ((ci + co) * es * P, f"wgrad {ci}->{co}: x, gy in")
        if k == 3 and train:
            return ((64 + 64) * es * P, "wgrad 3x3 (every 3x3 conv of the net but conv7 is 64->64; conv7: 1 launch in 17)")
        return (None, "mixed")                         # one instantiation serves several channel shapes
    table = {"cac_apply_c8_kernel": 6 * 64 * es * P + 4 * P, "cac_stats_c8_kernel": 128 * es * P + 8 * P,
             "cac_apply_kernel": 6 * 64 * 4 * P + 4 * P, "cac_stats_kernel": 128 * 4 * P + 8 * P,
             "head_c8_kernel": 64 * es * P + 8 * P, "head_kernel": 64 * 4 * P + 8 * P,
             "stem_c8_kernel": 64 * es * P + 4 * P, "stem_kernel": 64 * 4 * P + 4 * P,
             "conv1x1_c8_kernel<C8Bf16, 64, 128>": (64 + 128 + 128) * es * P,
             # per block: g_out, pre in; g_pre, g_in out (2 x 64 channels each) = 8 x 64; blocks 3..0 also READ the running
             # g_in (accumulate): 10 x 64 -- a training step launches 1 + 4 of them: 9.6 x 64 on average
             "cac_bwd_apply_c8_kernel": 9.6 * 64 * es * P, "cac_bwd_reduce_c8_kernel": 4 * 64 * es * P,
             # fused pass A: g_out, pre in (4 x 64) + sp, pooled max in, g_z, argch out (4 B per pixel each).  dL/d(inputs) is
             # no longer touched here (autograd.SUM_IN_DGRAD: the last dgrad convs add it; before: + 3.6 x 64)
             "cac_bwd_reduce_acc_c8_kernel": 4 * 64 * es * P + 16 * P}
    for k, v in table.items():
        if name.startswith(k):
            return (v, "single role")
    return (None, None)


def main():
    fd, wd, bd, out = sys.argv[1:5]
    B, H, W = (int(v) for v in sys.argv[5:8]) if len(sys.argv) >= 8 else (32, 480, 640)
    mode = sys.argv[8] if len(sys.argv) >= 9 else "fwd"
    es = int(sys.argv[9]) if len(sys.argv) >= 10 else 2
    assert mode in ("fwd", "train") and es in (2, 4)
    P = B * H * W
    unit = 1024.0
    fe, wr = load(fd, {"FETCH_SIZE"}), load(wd, {"WRITE_SIZE"})
    bu = load(bd, {"GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"}) if bd != "-" else {}
    ks = {}
    for k, rows in fe.items():
        n = len(rows)
        fb = 2.0 * unit * sum(r["FETCH_SIZE"] for r in rows) / n
        e = {"launches": n, "avg_ms_pmc_pass": sum(r["ns"] for r in rows) / n / 1e6, "fetch_bytes_corrected": fb}
        w = wr.get(k, [])
        if len(w) != n:
            e["pass_mismatch"] = f"FETCH pass {n} launches, WRITE pass {len(w)}"
        wb = unit * sum(r["WRITE_SIZE"] for r in w) / len(w) if w else 0.0
        e["write_bytes"] = wb
        e["hbm_bytes_per_launch"] = fb + wb
        ab, role = alg_bytes(k, P, es, mode)
        if role:
            e["roles"] = role
        if ab and "pass_mismatch" not in e:
            e["alg_bytes_per_launch"] = ab
            e["traffic_over_alg"] = (fb + wb) / ab
        b = bu.get(k, [])
        if b:
            if len(b) != n:
                e["busy_pass_mismatch"] = f"FETCH pass {n} launches, busy pass {len(b)}"
            good, rej = [], 0
            for r in b:
                if "GRBM_GUI_ACTIVE" not in r or "SQ_VALU_MFMA_BUSY_CYCLES" not in r or r["ns"] <= 0:
                    rej += 1
                    continue
                xcd_cycles = r["GRBM_GUI_ACTIVE"] / 8.0
                clock = xcd_cycles / r["ns"]
                busy = r["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * xcd_cycles) if xcd_cycles > 0 else 2.0
                # a launch shorter than 0.1 ms is dominated by the counters' start / stop skew: not judged, not averaged
                if r["ns"] < 1e5:
                    continue
                if not (0.5 <= clock <= 3.0) or busy > 1.0:
                    rej += 1
                    continue
                good.append((clock, busy, r["ns"]))
            e["busy_pass_rejected"] = rej
            if good and rej <= 0.25 * len(b):
                tns = sum(g[2] for g in good)
                e["clock_ghz"] = sum(g[0] * g[2] for g in good) / tns          # time-weighted = total cycles / total time
                e["mfma_busy_frac"] = sum(g[1] * g[0] * g[2] for g in good) / sum(g[0] * g[2] for g in good)
                e["avg_ms_busy_pass"] = tns / len(good) / 1e6
                assert 0.5 <= e["clock_ghz"] <= 3.0 and e["mfma_busy_frac"] <= 1.0
            elif rej > 0.25 * len(b):
                e["suspect"] = True
        ks[k] = e
    doc = {"shape": {"B": B, "H": H, "W": W}, "mode": mode, "esize": es,
           "method": "separate rocprofv3 --pmc passes of the same bench command, joined per dispatch index; FETCH_SIZE x2 "
                     "(gfx950: 128-B requests tallied at 64 B), WRITE_SIZE exact, both KB; clock = GRBM_GUI_ACTIVE / 8 / "
                     "time; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x per-XCD cycles), per dispatch, "
                     "time-weighted; dispatches outside 0.5-3 GHz or busy > 1 rejected.  Profiled passes clock lower than "
                     "un-profiled runs: compare ratios, not milliseconds.",
           "kernels": dict(sorted(ks.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}
    # which build of the kernels these counters belong to: bench.py prints it beside roofline.traffic (`traffic_from_hash`),
    # so a ratio measured before a later kernel change is visible from the line
    try:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from codon_amd import _lib
        doc["lib_source_hash"] = _lib.build_info()["source_hash_built"]
    except Exception as e:           # noqa: BLE001
        doc["lib_source_hash"] = None
        print(f"pmc_report: library source hash unavailable ({type(e).__name__}: {e})", file=sys.stderr)
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{'kernel':72s} {'n':>4s} {'ms':>7s} {'GB':>7s} {'x alg':>6s} {'GHz':>5s} {'busy':>5s}")
    for k, v in list(doc["kernels"].items())[:20]:
        print(f"{k[:72]:72s} {v['launches']:4d} {v['avg_ms_pmc_pass']:7.3f} {v['hbm_bytes_per_launch'] / 1e9:7.3f} "
              f"{v.get('traffic_over_alg', float('nan')):6.2f} {v.get('clock_ghz', float('nan')):5.2f} "
              f"{v.get('mfma_busy_frac', float('nan')):5.2f}" + ("  SUSPECT" if v.get("suspect") else ""))


if __name__ == "__main__":
    main()
