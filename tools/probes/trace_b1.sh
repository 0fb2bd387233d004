#!/bin/bash
# Run ON THE GPU BOX: kernel trace + per-kernel statistics + one forward's timeline for a one-image-per-call forward
#   trace_b1.sh <tag> <fp16|fp32|bf16> <H> <W>     ->  gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_timeline.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
tag=$1; dt=$2; H=$3; W=$4
OUT=$ROOT/gpurun_out; mkdir -p $OUT; rm -rf $OUT/${tag}_trace
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/trace_b1.py $dt $H $W 30 > $OUT/${tag}_untraced.txt 2>&1 || exit 1
CODON_B1_GRAPH=1 python3 $ROOT/tools/trace_b1.py $dt $H $W 30 >> $OUT/${tag}_untraced.txt 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${tag}_trace -- python3 $ROOT/tools/trace_b1.py $dt $H $W 30 > $OUT/${tag}_trace.log 2>&1 || exit 1
f=$(find $OUT/${tag}_trace -name "*kernel_stats.csv" | head -1); test -n "$f" || exit 1; cp "$f" $OUT/${tag}_kernel_stats.csv
t=$(find $OUT/${tag}_trace -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $OUT/${tag}_timeline.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
stems = [i for i, r in enumerate(rows) if "stem" in r["Kernel_Name"]]
# one forward = from its first stem launch to the next forward's (round 6: both stems are ONE launch, codon_stem_pair_fwd;
# before: two stem launches per forward); take one in the middle
per = 1 if any("pair" in rows[i]["Kernel_Name"] for i in stems) else 2
k = stems[(len(stems) // (2 * per)) * per]
k2 = stems[(len(stems) // (2 * per)) * per + per]
t0 = int(rows[k]["Start_Timestamp"])
print(f"# one forward: {k2 - k} launches, {(int(rows[k2]['Start_Timestamp']) - t0) / 1e3:.1f} us from its first launch to the next forward's first launch")
print("# start_us end_us dur_us gap_before_us queue kernel")
prev_end = t0
busy = 0
for r in rows[k:k2]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev_end) / 1e3:8.1f} q{r.get('Queue_Id')} {r['Kernel_Name'][:110]}")
    prev_end = max(prev_end, e)
PY
find $OUT/${tag}_trace -name "*.csv" -size +4M -delete
cat $OUT/${tag}_untraced.txt; tail -2 $OUT/${tag}_trace.log
