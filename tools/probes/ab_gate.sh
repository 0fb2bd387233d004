#!/bin/bash
# Run ON THE GPU BOX: bf16 conv5x5 64->64 plain / gated / gated + emitting under library variants ($@ = tags of tools/probes/bin)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd $ROOT
for rep in 1 2; do for v in "$@"; do CODON_AMD_LIB=$ROOT/tools/probes/bin/libcodon_hip_$v.so python3 tools/probes/time_gated.py 2>/dev/null | tail -1 || exit 1; done; done
