"""Merge rocprofv3 PMC passes of ONE bench command into a per-kernel table (profiles/rNN_*_pmc.json):

  python tools/pmc_report.py <fetch_dir> <write_dir> <busy_dir|-> <out.json> [B H W]

  fetch_dir : rocprofv3 --kernel-trace --pmc FETCH_SIZE ...
  write_dir : rocprofv3 --kernel-trace --pmc WRITE_SIZE ...                       (separate pass: TCC has 4 slots)
  busy_dir  : rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES ...   ("-" to skip)

Corrections (MI355X_MICROARCH.md, HBM section): both sizes are reported in KB; FETCH_SIZE tallies 128-B requests at
64 B on gfx950, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores.  clock = GRBM_GUI_ACTIVE / 8 XCDs /
kernel time; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x per-XCD active cycles).  Algorithmic bytes are
attached for the kernels whose byte count is a closed form of (B, H, W) -- a ratio well above 1 means re-reads."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(d, counters):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0, 0.0]))   # kernel -> counter -> [launches, sum, ns]
    for f in files:
        for row in csv.DictReader(open(f)):
            c = row.get("Counter_Name")
            if c not in counters:
                continue
            name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").replace("codon::", "").strip()
            a = acc[name][c]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return acc


def alg_bytes(name, P, es=2):
    """Bytes the kernel must move once per launch (16-bit activations: es = 2), or None."""
    m = re.match(r"conv_c8_kernel<C8\w+, (\d), (\d+), (\d+), (true|false)(?:, \d+, (true|false))?>", name)
    if m:
        k, ci, co, fuse, gate = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4) == "true", m.group(5) == "true"
        if fuse:
            return (ci + 64) * es * P                     # inference: 128 in + 64 out (training also writes the 128-ch mid)
        return ((2 if gate else 1) * ci + co) * es * P    # gated staging reads pre AND inputs; dgrad epilogues add mask / accumulate reads
    m = re.match(r"conv_mfma_bf16_kernel<\w+, (\d), (\d+), (\d+)", name)
    if m:
        return (int(m.group(2)) + int(m.group(3))) * es * P
    table = {"cac_apply_c8_kernel": 6 * 64 * es * P + 4 * P, "cac_stats_c8_kernel": 128 * es * P + 8 * P,
             "cac_apply_kernel": 6 * 64 * es * P + 4 * P, "cac_stats_kernel": 128 * es * P + 8 * P,
             "head_c8_kernel": 64 * es * P + 8 * P, "head_kernel": 64 * es * P + 8 * P,
             "stem_c8_kernel": 64 * es * P + 4 * P, "stem_kernel": 64 * es * P + 4 * P,
             "conv1x1_c8_kernel<C8Bf16, 64, 128>": (64 + 128 + 128) * es * P,
             "cac_bwd_apply_c8_kernel": (2 + 2 + 2 + 2) * 64 * es * P, "cac_bwd_reduce_c8_kernel": 4 * 64 * es * P}
    for k, v in table.items():
        if name.startswith(k):
            return v
    return None


def main():
    fd, wd, bd, out = sys.argv[1:5]
    B, H, W = (int(v) for v in sys.argv[5:8]) if len(sys.argv) >= 8 else (32, 480, 640)
    P = B * H * W
    unit = 1024.0
    fe, wr = load(fd, {"FETCH_SIZE"}), load(wd, {"WRITE_SIZE"})
    bu = load(bd, {"GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"}) if bd != "-" else {}
    ks = {}
    for k in fe:
        f = fe[k]["FETCH_SIZE"]
        n = f[0]
        fb = 2.0 * unit * f[1] / n
        w = wr.get(k, {}).get("WRITE_SIZE")
        wb = unit * w[1] / w[0] if w and w[0] else 0.0
        e = {"launches": n, "avg_ms_pmc_pass": f[2] / n / 1e6, "fetch_bytes_corrected": fb, "write_bytes": wb,
             "hbm_bytes_per_launch": fb + wb}
        ab = alg_bytes(k, P)
        if ab:
            e["alg_bytes_per_launch"] = ab
            e["traffic_over_alg"] = (fb + wb) / ab
        if k in bu and "GRBM_GUI_ACTIVE" in bu[k] and "SQ_VALU_MFMA_BUSY_CYCLES" in bu[k]:
            g, m = bu[k]["GRBM_GUI_ACTIVE"], bu[k]["SQ_VALU_MFMA_BUSY_CYCLES"]
            ns = g[2] / g[0]
            xcd_cycles = g[1] / g[0] / 8.0
            e["clock_ghz"] = xcd_cycles / ns
            e["mfma_busy_frac"] = (m[1] / m[0]) / (1024.0 * xcd_cycles)
            e["avg_ms_busy_pass"] = ns / 1e6
        ks[k] = e
    doc = {"shape": {"B": B, "H": H, "W": W},
           "method": "separate rocprofv3 --pmc passes of the same bench command; FETCH_SIZE x2 (gfx950: 128-B requests tallied "
                     "at 64 B), WRITE_SIZE exact, both KB; clock = GRBM_GUI_ACTIVE / 8 / time; mfma_busy = "
                     "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x per-XCD cycles).  Profiled passes clock lower than "
                     "un-profiled runs: compare ratios, not milliseconds.",
           "kernels": dict(sorted(ks.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"]))}
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{'kernel':64s} {'n':>4s} {'ms':>7s} {'GB':>7s} {'x alg':>6s} {'GHz':>5s} {'busy':>5s}")
    for k, v in list(doc["kernels"].items())[:16]:
        print(f"{k[:64]:64s} {v['launches']:4d} {v['avg_ms_pmc_pass']:7.3f} {v['hbm_bytes_per_launch'] / 1e9:7.3f} "
              f"{v.get('traffic_over_alg', float('nan')):6.2f} {v.get('clock_ghz', float('nan')):5.2f} "
              f"{v.get('mfma_busy_frac', float('nan')):5.2f}")


if __name__ == "__main__":
    main()
