# same-box A/B of two builds of the 16-bit wgrad: tools/probes/bin/libcodon_hip_$1.so against the in-tree library
mkdir -p gpurun_out/w
for rep in 1 2; do
  for a in "$@" base; do
    if [ $a = base ]; then unset CODON_AMD_LIB; else export CODON_AMD_LIB=$PWD/tools/probes/bin/libcodon_hip_$a.so; fi
    echo "== $a" >> gpurun_out/w/ab.txt
    DATA=relu python tools/time_wgrad.py bf16 2>&1 | grep wgrad >> gpurun_out/w/ab.txt
  done
done
cat gpurun_out/w/ab.txt
