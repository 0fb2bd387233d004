set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python -m pytest tests/test_gpu_c8.py tests/test_gpu_backward.py tests/test_boundary.py -q -x -p no:cacheprovider > gpurun_out/r4i/t1.log 2>&1; echo "c8/backward/boundary rc=$?"; tail -3 gpurun_out/r4i/t1.log | cut -c1-300
for i in 1 2; do
python bench.py --mode train --dtype bf16 --steps 6 --warmup 2 > gpurun_out/r4i/train_sum$i.json 2> gpurun_out/r4i/train_sum$i.err; python -c "
import json; d=json.load(open('gpurun_out/r4i/train_sum$i.json')); print('bf16 train (sum4)', d['ms_per_step'], d['peak_mem_gb'])"
done
