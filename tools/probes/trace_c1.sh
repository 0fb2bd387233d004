ROOT=$(pwd); OUT=$ROOT/gpurun_out/c1t; mkdir -p $OUT; rm -rf $OUT/*
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $ROOT/tools/time_c1.py > $OUT/log.txt 2>&1
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a window in the middle of the eager loop
mid = next(i for i in range(len(rows) // 3, len(rows)) if 'stem' in rows[i]['Kernel_Name'])
t0 = int(rows[mid]["Start_Timestamp"])
print(rows[0].keys())
for r in rows[mid:mid + int(__import__('os').environ.get('NROWS', '40'))]:
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f} q{r.get("Queue_Id")} s{r.get("Stream_Id", "")} {r["Kernel_Name"][:70]}')
PY
tail -2 $OUT/log.txt
find $OUT/t -name "*.csv" -delete
