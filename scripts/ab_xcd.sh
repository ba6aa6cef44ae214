cd $GRAFT_REPO_ROOT
V=build/variants/noxcd/libhimo_amd.so
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-extra-precisions > gpurun_out/ab_xcd_$i.json 2>/dev/null
  HIMO_AMD_LIB=$PWD/$V python bench.py --no-cpu-baseline --no-extra-precisions > gpurun_out/ab_noxcd_$i.json 2>/dev/null
done
bash scripts/trace_layers.sh trace_xcd > gpurun_out/trace_xcd.txt 2>&1
HIMO_AMD_LIB=$PWD/$V bash scripts/trace_layers.sh trace_noxcd > gpurun_out/trace_noxcd.txt 2>&1
