# Run ON THE GPU BOX: HBM bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, KB) of the 5x5 wgrad launches for the
# in-tree build and variants:   pmc_wgrad_traffic.sh <tag> ...
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/pwt
cd /tmp && export TMPDIR=/tmp
for a in base "$@"; do
  if [ $a = base ]; then unset CODON_AMD_LIB; else export CODON_AMD_LIB=$ROOT/tools/probes/bin/libcodon_hip_$a.so; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    DATA=relu rocprofv3 --kernel-trace --pmc $c --output-format csv -d $ROOT/gpurun_out/pwt/${a}_$c -- python3 $ROOT/tools/time_wgrad.py bf16 ${CASE:-0} > $ROOT/gpurun_out/pwt/${a}_$c.log 2>&1
  done
  python3 - $ROOT/gpurun_out/pwt/$a $a <<'PY'
import csv, glob, sys
d, tag = sys.argv[1], sys.argv[2]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    n = v = 0.0
    for f in glob.glob(d + "_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_wgrad_c8_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                v += float(r["Counter_Value"]); n += 1
    # one row per (dispatch, instance): count dispatches by distinct ids
    ids = set()
    for f in glob.glob(d + "_" + c + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_wgrad_c8_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                ids.add(r["Dispatch_Id"])
    tot[c] = v * 1024 / max(1, len(ids))
gb = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / 1e9
alg = 32 * 480 * 640 * (128 + 128) * 2 / 1e9 if True else 0
print(f"{tag}: {gb:.3f} GB per launch (fetch x2 {2 * tot['FETCH_SIZE'] / 1e9:.3f} + write {tot['WRITE_SIZE'] / 1e9:.3f})")
PY
  find $ROOT/gpurun_out/pwt -name "*.csv" -size +2M -delete
done
