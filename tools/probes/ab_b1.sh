#!/bin/bash
# Run ON THE GPU BOX: one-image-per-call latency under the round-5 schedule switches (same box, alternating)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
run() { echo "== $DT ${H}x$W $*"; env "$@" CODON_B1_GRAPH=1 python3 tools/trace_b1.py $DT $H $W 50 2>/dev/null | grep "ms/forward" || exit 1; }
for rep in 1 2; do
  DT=fp32; H=128; W=128
  run CODON_CAC_TAIL=0 CODON_PAIR_MAX32=0
  run CODON_CAC_TAIL=1 CODON_PAIR_MAX32=0
  run CODON_CAC_TAIL=1 CODON_PAIR_MAX32=383
  DT=fp16; H=370; W=463
  run CODON_CAC_TAIL=0 CODON_PAIR_MAX16=0
  run CODON_CAC_TAIL=1 CODON_PAIR_MAX16=4096
  DT=fp16; H=128; W=128
  run CODON_CAC_TAIL=1 CODON_PAIR_MAX16=0
  run CODON_CAC_TAIL=1 CODON_PAIR_MAX16=4096
done
