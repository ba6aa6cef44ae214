#!/bin/bash
# HBM bytes per launch of the head's gate weight gradients (scripts/exp_linear_wgrad.py): FETCH_SIZE (KiB, x2 on gfx950 per MI355X_MICROARCH.md's
# HBM section) and WRITE_SIZE in separate rocprofv3 --pmc passes.  usage (GPU box, repo root): bash scripts/exp_linear_wgrad_pmc.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/lwgpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python $R/scripts/exp_linear_wgrad.py > $OUT/$c.log 2>&1
done
cd $R
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if r["Counter_Name"] == c and ("wgrad_full_split_kernel" in n or "wgrad_partial_split_kernel" in n):
                acc[(n.split("(")[0].replace("void ", "").replace("himo::", ""), r["Grid_Size"])][c].append(float(r["Counter_Value"]))
print("kernel, grid size: HBM bytes per launch (FETCH_SIZE KiB x 1024 x 2 read, WRITE_SIZE KiB x 1024 written), launches")
for k in sorted(acc):
    f, w = acc[k]["FETCH_SIZE"], acc[k]["WRITE_SIZE"]
    if f and w:
        print(f"  {k[0]:40s} grid {k[1]:>8s}: read {sum(f) / len(f) * 2048 / 1e6:8.1f} MB  written {sum(w) / len(w) * 1024 / 1e6:7.1f} MB   ({len(f)} launches)")
PY
