# same-box A/B of builds of the 16-bit wgrad: tools/probes/bin/libcodon_hip_<tag>.so ... against the in-tree library
#   bash tools/probes/ab_wgrad.sh <case index of tools/time_wgrad.py | all> <tag> ...
mkdir -p gpurun_out/w
rm -f gpurun_out/w/ab.txt
c=$1; shift
[ "$c" = all ] && c=""
for rep in 1 2; do
  for a in "$@" base; do
    if [ $a = base ]; then unset CODON_AMD_LIB; else export CODON_AMD_LIB=$PWD/tools/probes/bin/libcodon_hip_$a.so; fi
    echo "== $a" >> gpurun_out/w/ab.txt
    DATA=relu python tools/time_wgrad.py bf16 $c 2>&1 | grep wgrad >> gpurun_out/w/ab.txt
  done
done
cat gpurun_out/w/ab.txt
