# clock + matrix-pipe busy of the 5x5 wgrad for the in-tree build and variants: GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/pw
cd /tmp && export TMPDIR=/tmp
for a in base "$@"; do
  if [ $a = base ]; then unset CODON_AMD_LIB; else export CODON_AMD_LIB=$ROOT/tools/probes/bin/libcodon_hip_$a.so; fi
  DATA=relu rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $ROOT/gpurun_out/pw/$a -- python3 $ROOT/tools/time_wgrad.py bf16 0 > $ROOT/gpurun_out/pw/$a.log 2>&1
  python3 - $ROOT/gpurun_out/pw/$a $a <<'PY'
import csv, glob, sys, collections
d, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wgrad_c8_kernel" not in r["Kernel_Name"]:
            continue
        a = acc[r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
g, m = acc["GRBM_GUI_ACTIVE"], acc["SQ_VALU_MFMA_BUSY_CYCLES"]
ns = g[2] / g[0]; cyc = g[1] / g[0] / 8
print(f"{tag}: {ns/1e6:.3f} ms  clock {cyc/ns:.3f} GHz  mfma busy {m[1]/m[0]/(1024*cyc):.3f}")
PY
done
