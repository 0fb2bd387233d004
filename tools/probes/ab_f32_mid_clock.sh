#!/bin/bash
# Run ON THE GPU BOX: duration and shader clock (GRBM_GUI_ACTIVE / 8 XCDs / time) of every conv5x5-128 launch of the fp32 one-image
# forward at 370 x 463, round-4 tree vs current
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/f32clk; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for t in r04 r05; do
  if [ $t = r04 ]; then TB=$ROOT/ab/r04/trace_b1.py; else TB=$ROOT/tools/trace_b1.py; fi
  rm -rf $OUT/$t
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/$t -- python3 $TB fp32 370 463 6 > $OUT/$t.log 2>&1
  python3 - $OUT/$t $t <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
per = collections.OrderedDict()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "5, 128, 128" not in r["Kernel_Name"]: continue
        k = int(r["Dispatch_Id"])
        e = per.setdefault(k, {"ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = [per[k] for k in sorted(per)][-26:]
print(tag, " ".join(f"{r['ns']/1e6:.2f}ms@{r.get('GRBM_GUI_ACTIVE',0)/8/r['ns']:.2f}GHz/b{r.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(1024*r.get('GRBM_GUI_ACTIVE',1)/8):.2f}" for r in rows))
PY
  find $OUT/$t -name "*.csv" -size +1M -delete
done
