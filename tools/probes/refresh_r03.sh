set -e -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out; mkdir -p $OUT
bash $ROOT/tools/profile_16bit.sh r03 2>&1 | grep -E "^==|conv_c8|wgrad|cac|done" | cut -c1-130
cd $ROOT
echo "== bench lines"
python bench.py > $OUT/r03_bench_default.json 2> $OUT/r03_bench_default.err
python bench.py --dtype bf16 > $OUT/r03_bench_bf16.json 2> $OUT/r03_bench_bf16.err
python bench.py --mode train --dtype bf16 --steps 10 --warmup 2 > $OUT/r03_train_bf16.json 2> $OUT/r03_train_bf16.err
echo finished
