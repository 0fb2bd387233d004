# timing ablations of conv_c8_kernel<3,64,64>: apply tools/probes/conv3_ablation.patch, build variants with
#   tools/ab_build.sh c<mask> conv_c8.hip -DABL=<mask>, revert the patch, then run this on the GPU box
mkdir -p gpurun_out/abl; rm -f gpurun_out/abl/c3.txt
for rep in 1 2; do
for a in base "$@"; do
  if [ $a = base ]; then unset CODON_AMD_LIB; else export CODON_AMD_LIB=$PWD/tools/probes/bin/libcodon_hip_$a.so; fi
  echo "== $a" >> gpurun_out/abl/c3.txt
  for i in 1 2 3; do DATA=relu python tools/time_conv.py bf16 2 2>&1 | grep "^conv" >> gpurun_out/abl/c3.txt; done
done
done
cat gpurun_out/abl/c3.txt
