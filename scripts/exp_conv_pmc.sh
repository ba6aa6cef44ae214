#!/bin/bash
# PMC look at the split-precision convolution kernel on the network's layer shapes (one precision per run).
# usage (on the GPU box, repo root): bash scripts/exp_conv_pmc.sh f16x2
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
PREC=${1:-f16x2}
OUT=$R/gpurun_out/convpmc_$PREC
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $R/scripts/exp_conv.py $PREC ${2:-0} > $OUT/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "conv_bf16x3_kernel" not in name and "conv_mfma_kernel" not in name and "conv3_split_kernel" not in name: continue
        key = (name[:60], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
for key in sorted(agg):
    print(key)
    for c, v in sorted(agg[key].items()):
        print(f"    {c:32s} {v / cnt[(key, c)]:16.0f}")
PY
