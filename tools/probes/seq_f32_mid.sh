#!/bin/bash
# Run ON THE GPU BOX: the kernel sequence (start, duration, gap) of the LAST fp32 one-image forward at 370 x 463, per tree
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/f32seq; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  TB=$ROOT/ab/$t/trace_b1.py
  rm -rf $OUT/$t
  rocprofv3 --kernel-trace --output-format csv -d $OUT/$t -- python3 $TB fp32 370 463 8 > $OUT/$t.log 2>&1
  python3 - $OUT/$t $t > $OUT/${t}_seq.txt <<'PY'
import csv, glob, sys, re
d, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one forward = the kernels between two consecutive launches of the first kernel of the forward; take the last full one
n = len(rows)
per = n // 13                      # 5 warm-up + 8 timed forwards
last = rows[n - per:]
t0 = int(last[0]["Start_Timestamp"]); prev_end = t0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"^void codon::", "", r["Kernel_Name"])[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:6.1f}  q{r.get('Queue_Id', '?')}  {name}")
    prev_end = e
PY
  rm -rf $OUT/$t
done
