#!/bin/bash
# Timing-only variant: the chained fp32 conv5x5-128 on 16 x 32 tiles, ONE 4-wave workgroup per CU (PSEG = 4: 16 accumulator
# tiles per wave) when CODON_PROBE_PSEG4 is set  ->  tools/probes/bin/libcodon_hip_pseg4.so
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
T=$(mktemp -d)
cp -r "$ROOT/codon_amd/csrc" "$T/csrc"
python3 - "$T/csrc/conv_mfma_f32.hip" <<'PY'
import sys
f = sys.argv[1]
s = open(f).read()
a = "__global__ __launch_bounds__(NW * 64, 2) void conv_mfma_f32_kernel(const ConvParams p) {"
assert a in s
s = s.replace(a, "__global__ __launch_bounds__(NW * 64, PSEG >= 4 ? 1 : 2) void conv_mfma_f32_kernel(const ConvParams p) {", 1)
b = "  return launch_or_hold_f32<5, 128, 128, 2, true, false, NWC>(p, false, stream);\n}"
assert b in s
s = s.replace(b, """  if (getenv("CODON_PROBE_PSEG4")) {
    p.tiles_y = (d->height + 15) / 16;
    p.nblk = (int)((long)p.tiles_x * p.tiles_y * d->batch);
    return launch_single_f32<5, 128, 128, 4, true, false, NWC, false>(&p, stream);
  }
  return launch_or_hold_f32<5, 128, 128, 2, true, false, NWC>(p, false, stream);
}""", 1)
open(f, "w").write("#include <cstdlib>\n" + s)
PY
mkdir -p "$ROOT/tools/probes/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I"$ROOT/include" -I"$T/csrc" -Wall -Wno-unused-function -c "$T/csrc/conv_mfma_f32.hip" -o "$T/ab_conv_mfma_f32.o"
objs=$(ls "$ROOT"/codon_amd/csrc/build/*.o | grep -v "build/ab_" | grep -v "build/conv_mfma_f32.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/probes/bin/libcodon_hip_pseg4.so" $objs "$T/ab_conv_mfma_f32.o"
cd "$T" && objcopy -O binary --only-section=.hip_fatbin ab_conv_mfma_f32.o f.fat && /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=f.fat --output=f.co --unbundle && /opt/rocm/lib/llvm/bin/llvm-readelf --notes f.co | grep -E "\.name:|group_segment_fixed_size|\.vgpr_count|\.agpr_count|vgpr_spill" | paste - - - - - | grep "ILi5ELi128ELi128ELi4E" | sed 's/ \+/ /g'
rm -rf "$T"
echo "$ROOT/tools/probes/bin/libcodon_hip_pseg4.so"
