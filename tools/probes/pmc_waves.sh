#!/bin/bash
# Run ON THE GPU BOX: wave-state counters of one bf16 training step (where do waves spend their cycles: parked at
# s_waitcnt / barrier, stalled on issue = waiting for a pipe, or issuing).  Usage: pmc_waves.sh [tag] [bench.py args...]
# (default: one bf16 training step).  Output: stdout (the caller redirects it) + gpurun_out/pmc_waves_<tag>/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
TAG=${1:-train}; shift || true
ARGS=${@:---mode train --dtype bf16 --steps 1 --warmup 0 --no-cpu-baseline}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_waves_$TAG -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_waves_$TAG.log 2>&1 || echo failed
python3 - <<PY
import csv, glob, re, collections
fs = sorted(glob.glob("$OUT/pmc_waves_$TAG/**/*counter_collection.csv", recursive=True))[-1:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
for f in fs:
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("codon::", "")
        a = acc[n][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = []
for n, cs in acc.items():
    if "SQ_WAVE_CYCLES" not in cs or cs["SQ_WAVE_CYCLES"][1] == 0: continue
    wc = cs["SQ_WAVE_CYCLES"][1]
    rows.append((cs["SQ_WAVE_CYCLES"][2], n, cs["SQ_WAVE_CYCLES"][0], cs["SQ_WAVE_CYCLES"][2] / cs["SQ_WAVE_CYCLES"][0] / 1e6,
                 cs["SQ_WAIT_ANY"][1] / wc, cs["SQ_WAIT_INST_ANY"][1] / wc, cs["SQ_ACTIVE_INST_ANY"][1] / wc, cs["SQ_WAIT_INST_LDS"][1] / wc))
print(f"{'kernel':78s} {'n':>4s} {'ms':>7s} {'parked':>7s} {'pipe':>7s} {'issue':>7s} {'lds':>6s}")
for t, n, c, ms, a, b, d, e in sorted(rows, reverse=True)[:22]:
    print(f"{n[:78]:78s} {c:4d} {ms:7.3f} {a:7.2f} {b:7.2f} {d:7.2f} {e:6.2f}")
PY
