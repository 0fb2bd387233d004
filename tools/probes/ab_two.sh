#!/bin/bash
# Run ON THE GPU BOX: same-box alternating A/B of two library builds (tools/ab_build.sh tags) on one timing script
#   ab_two.sh <tagA> <tagB> <script.py> [args...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
B=$ROOT/tools/probes/bin
a=$1; b=$2; s=$3; shift 3
for rep in 1 2 3; do for v in $a $b; do echo "== $v"; CODON_AMD_LIB=$B/libcodon_hip_$v.so python3 $s "$@" 2>&1 | grep -v "amdgpu.ids\|CODON_AMD_LIB" || exit 1; done; done
