# timing ablations of conv_wgrad_c8_kernel<5>: apply tools/probes/wgrad_ablation.patch, build variants with
#   tools/ab_build.sh n<mask> conv_wgrad_c8.hip -DABL=<mask>, revert the patch, then run this on the GPU box
mkdir -p gpurun_out/abl
rm -f gpurun_out/abl/out.txt
for a in base "$@"; do
  if [ $a = base ]; then unset CODON_AMD_LIB; else export CODON_AMD_LIB=$PWD/tools/probes/bin/libcodon_hip_$a.so; fi
  echo "== $a" >> gpurun_out/abl/out.txt
  DATA=relu python tools/time_wgrad.py bf16 0 2>&1 | grep wgrad >> gpurun_out/abl/out.txt
done
cat gpurun_out/abl/out.txt
