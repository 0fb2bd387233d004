set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
python -m pytest tests/test_gpu_c8.py -q -x -p no:cacheprovider -k "fused_cac or training_schedules" > gpurun_out/r4f/t1.log 2>&1; echo "fused tests rc=$?"; tail -15 gpurun_out/r4f/t1.log | cut -c1-300
python -m pytest tests/test_gpu_backward.py tests/test_gpu_dist.py -q -x -p no:cacheprovider -s > gpurun_out/r4f/t2.log 2>&1; echo "backward+dist rc=$?"; grep -E "ranks\]|passed|failed" gpurun_out/r4f/t2.log | cut -c1-400
for f in 1 0 1 0; do
CODON_FUSED_CAC_BWD=$f python bench.py --mode train --dtype bf16 --steps 6 --warmup 2 > gpurun_out/r4f/train_f$f.json 2> gpurun_out/r4f/train_f$f.err; python -c "
import json; d=json.load(open('gpurun_out/r4f/train_f$f.json')); print('bf16 train fused_cac=$f', d['ms_per_step'], d['peak_mem_gb'])"
done
