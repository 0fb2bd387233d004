#!/bin/bash
# Run ON THE GPU BOX: same-box alternating A/B of two library builds on tools/time_conv.py
#   ab_two.sh <tagA> <tagB> <mode> [case]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
B=$ROOT/tools/probes/bin
for rep in 1 2 3; do for v in $1 $2; do echo "== $v"; CODON_AMD_LIB=$B/libcodon_hip_$v.so python3 tools/time_conv.py $3 $4 2>&1 | grep "^conv" || exit 1; done; done
