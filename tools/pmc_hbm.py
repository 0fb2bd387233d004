"""Post-process two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command, as
MI355X_MICROARCH.md's HBM section prescribes) into profiles/rNN_*_pmc_hbm.json: HBM bytes per launch per kernel.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
  python tools/pmc_hbm.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_fwd_b32_480x640_pmc_hbm.json

gfx950 corrections: FETCH_SIZE and WRITE_SIZE are reported in KiB-like units of 1024 B? -- no: rocprofv3 reports both
in KB (x1024 to bytes); FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled; WRITE_SIZE is exact.  The
calibration check printed at the end compares the HBM-bound cac_apply kernel with its algorithmic byte count."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, f"no counter_collection.csv under {d}"
    acc = defaultdict(lambda: [0, 0.0, 0.0])       # kernel -> [launches, sum counter, sum ns]
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = re.sub(r"\(.*$", "", row["Kernel_Name"]).replace("void ", "").strip()
            a = acc[name]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            a[2] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    unit = 1024.0                                  # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KB
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    ks = {}
    for k in fe:
        n = fe[k][0]
        fb = 2.0 * unit * fe[k][1] / n             # gfx950: 128-B requests tallied at 64 B
        wb = unit * wr[k][1] / max(wr[k][0], 1) if k in wr else 0.0
        ks[k] = {"launches": n, "avg_ms_fetch_pass": fe[k][2] / n / 1e6,
                 "avg_ms_write_pass": (wr[k][2] / max(wr[k][0], 1) / 1e6) if k in wr else None,
                 "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes_per_launch": fb + wb}
    doc = {"command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 bench.py "
                      "--steps 1 --warmup 0 --no-cpu-baseline",
           "note": "FETCH_SIZE on gfx950 counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): corrected x2; "
                   "WRITE_SIZE exact.  Per launch = counter summed over launches / launches.",
           "kernels": ks}
    json.dump(doc, open(out, "w"), indent=1)
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:12]:
        print(f"{k[:70]:70s} n={v['launches']:3d} fetch {v['fetch_bytes_corrected']/1e9:7.3f} GB  write {v['write_bytes']/1e9:7.3f} GB")


if __name__ == "__main__":
    main()
