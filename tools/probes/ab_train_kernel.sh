#!/bin/bash
# Run ON THE GPU BOX: one kernel's average duration inside the bf16 training step, in-tree library vs tools/probes/bin variants
#   ab_train_kernel.sh "<kernel name substring>" <tag> ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out/abtk; mkdir -p $OUT
key=$1; shift
cd /tmp && export TMPDIR=/tmp
for a in "$@" base "$@" base; do
  if [ $a = base ]; then unset CODON_AMD_LIB; else export CODON_AMD_LIB=$ROOT/tools/probes/bin/libcodon_hip_$a.so; fi
  rm -rf $OUT/$a
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$a -- python3 $ROOT/bench.py --mode train --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$a.json 2> $OUT/$a.err
  f=$(find $OUT/$a -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$key" "$a" $OUT/$a.json <<'PY'
import csv, json, sys
f, key, tag, js = sys.argv[1:5]
ms = json.load(open(js))["ms_per_step"]
for r in csv.DictReader(open(f)):
    if key in r["Name"]:
        print(f"{tag:8s} step {ms:7.2f} ms (under the profiler)   {r['Calls']:>4} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:90]}")
PY
  find $OUT/$a -name "*kernel_trace.csv" -delete
done
