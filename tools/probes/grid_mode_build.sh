#!/bin/bash
# Timing-only variant of libcodon_hip.so whose fp32 conv launches take their tile height / CU sharing from the environment:
#   CODON_PROBE_SMALL=1   every launch takes 4 x 32 tiles (PSEG = 1);  =0 never;  unset: the product rule
#   CODON_PROBE_NOSOLO=1  PSEG = 1 launches without the one-workgroup-per-CU LDS padding
# -> tools/probes/bin/libcodon_hip_gridmode.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
cp -r "$ROOT/codon_amd/csrc" "$T/csrc"
python3 - "$T/csrc/conv_mfma_f32.hip" <<'PY'
import sys
f = sys.argv[1]
s = open(f).read()
a = "static bool small_grid(const codon_conv_desc* d) {\n"
assert a in s
s = s.replace(a, a + '  if (const char* e_ = getenv("CODON_PROBE_SMALL")) return e_[0] == \'1\';\n', 1)
b = "    if (small) {\n      if (pair_hold("
assert b in s
s = s.replace(b, '    if (small && !getenv("CODON_PROBE_NOSOLO")) {\n      if (pair_hold(', 1)
c = "  if (small && !pair_recorder() && nblk <= CSPLIT_MAX_BLOCKS) {"
assert c in s
s = s.replace(c, '  if (small && !pair_recorder() && nblk <= CSPLIT_MAX_BLOCKS && !getenv("CODON_PROBE_SMALL")) {', 1)
d = "      if (!pair_recorder() && nblk4 <= CSPLIT_MAX_BLOCKS) return"
assert d in s
s = s.replace(d, '      if (!pair_recorder() && nblk4 <= CSPLIT_MAX_BLOCKS && !getenv("CODON_PROBE_SMALL")) return', 1)
open(f, "w").write("#include <cstdlib>\n" + s)
PY
mkdir -p "$ROOT/tools/probes/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I"$ROOT/include" -I"$T/csrc" -Wall -Wno-unused-function -c "$T/csrc/conv_mfma_f32.hip" -o "$T/ab_conv_mfma_f32.o"
objs=$(ls "$ROOT"/codon_amd/csrc/build/*.o | grep -v "build/ab_" | grep -v "build/conv_mfma_f32.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/probes/bin/libcodon_hip_gridmode.so" $objs "$T/ab_conv_mfma_f32.o"
rm -rf "$T"
echo "$ROOT/tools/probes/bin/libcodon_hip_gridmode.so"
