#!/bin/bash
# Run ON THE GPU BOX: wave-state / LDS / L2 counters of the bf16 conv kernels stand-alone (tools/time_conv.py), one
# rocprofv3 --pmc pass per counter group (program directly after "--").  Output: gpurun_out/pmc_conv3_*.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DATA=relu
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" \
           "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_conv3_$i -- python3 $ROOT/tools/time_conv.py bf16 > $OUT/pmc_conv3_$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - <<PY
import csv, glob, re, collections
for i in range(1, 5):
    fs = glob.glob("$OUT/pmc_conv3_%d/**/*counter_collection.csv" % i, recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
    for f in fs:
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*$", "", r["Kernel_Name"]).replace("void ", "").replace("codon::", "")
            if "conv_c8_kernel" not in n: continue
            a = acc[n][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    for n, cs in acc.items():
        print(n[:70], {c: (round(v[1] / v[0]), round(v[2] / v[0] / 1e3)) for c, v in cs.items()})
PY
