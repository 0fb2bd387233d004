set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
python -m pytest tests/test_gpu_c8.py tests/test_gpu_backward.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider > gpurun_out/r4k/t1.log 2>&1; echo "c8/backward/kernels rc=$?"; tail -3 gpurun_out/r4k/t1.log | cut -c1-300
for f in 1 0 1 0; do
CODON_MASK_IN_EPILOGUE=$f python bench.py --mode train --dtype bf16 --steps 6 --warmup 2 > gpurun_out/r4k/train_m$f.json 2> gpurun_out/r4k/train_m$f.err; python -c "
import json; d=json.load(open('gpurun_out/r4k/train_m$f.json')); print('bf16 train mask_in_epilogue=$f', d['ms_per_step'])"
done
