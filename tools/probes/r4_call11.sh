set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
for f in 1 0 1 0; do
CODON_SUM_GFUSE=$f python bench.py --mode train --dtype bf16 --steps 6 --warmup 2 > gpurun_out/r4j/train_s$f.json 2> gpurun_out/r4j/train_s$f.err; python -c "
import json; d=json.load(open('gpurun_out/r4j/train_s$f.json')); print('bf16 train sum_gfuse=$f', d['ms_per_step'], d['peak_mem_gb'])"
done
