#!/bin/bash
# Effective shader clock (GRBM_GUI_ACTIVE / kernel duration) and matrix-pipe busy share of the 3x3 layer kernels for one or
# more library variants: is a faster variant faster in CYCLES, or only in CLOCK (power)?  usage (GPU box):
#   bash scripts/exp_clock_pmc.sh default nostore ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  OUT=$R/gpurun_out/clockpmc_$v
  rm -rf $OUT; mkdir -p $OUT
  if [ "$v" = default ]; then unset HIMO_AMD_LIB; else export HIMO_AMD_LIB=$R/build/variants/$v/libhimo_amd.so; fi
  timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT -o pmc -- python $R/scripts/exp_layers.py 16 > $OUT/run.log 2>&1
  python - <<PY
import csv, glob, collections
dur = {}
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv3_presplit_kernel" not in r["Kernel_Name"]: continue
        d = dur.get(r["Dispatch_Id"])
        if d is None: continue
        key = (r["Kernel_Name"].replace("void himo::", "")[:44], r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["ns"].append(d[0])
print("variant $v")
for key, c in sorted(agg.items()):
    n = len(c["GRBM_GUI_ACTIVE"])
    if n < 4: continue
    gui = sum(c["GRBM_GUI_ACTIVE"][-4:]) / 4
    ns = sum(c["ns"][-4 * 3:]) / len(c["ns"][-4 * 3:])
    mf = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"][-4:]) / 4
    print(f"  {key[0]:44s} grid {key[1]:>9s}: {ns/1e3:8.1f} us  gui cycles/XCD {gui/8/1e3:8.1f} k  clock {gui/8/ns:5.2f} GHz  mfma busy {mf/1024/(gui/8):5.1%}")
PY
done
