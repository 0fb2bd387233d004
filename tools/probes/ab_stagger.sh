#!/bin/bash
# Run ON THE GPU BOX: does phase-shifting the co-resident workgroups of the fp32 64-cout convs recover their per-tile bubbles?
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export CODON_AMD_LIB=$ROOT/tools/probes/bin/libcodon_hip_stag.so
cd $ROOT
for u in 0 5 10 15 0 10; do echo "3x3 occ4 units=$u"; CODON_STAG_UNITS=$u CODON_STAG_OCC=4 python3 tools/time_conv.py f32 2 || exit 1; done
for u in 0 13 26 0 26; do echo "5x5 occ3 units=$u"; CODON_STAG_UNITS=$u CODON_STAG_OCC=3 python3 tools/time_conv.py f32 1 || exit 1; done
