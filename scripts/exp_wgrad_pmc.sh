#!/bin/bash
# PMC look at the 3x3 weight-gradient kernel (split-bf16 operands) on the training step's layer shapes (scripts/exp_wgrad.py).
# usage (on the GPU box, repo root): bash scripts/exp_wgrad_pmc.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/wgradpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $R/scripts/exp_wgrad.py > $OUT/p$i.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "conv_wgrad_split_kernel" not in name: continue
        key = (name[:40], r["Grid_Size"])
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(key, r["Counter_Name"])] += 1
for key in sorted(agg):
    print(key)
    for c, v in sorted(agg[key].items()):
        print(f"    {c:32s} {v / cnt[(key, c)]:16.0f}")
PY
rm -rf $OUT/p*/
