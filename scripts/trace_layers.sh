#!/bin/bash
# per-(kernel, grid) average durations of one pipeline bench run: which layer costs what inside the network
# usage (GPU box, repo root): [HIMO_AMD_LIB=...] bash scripts/trace_layers.sh [out-name] [extra bench flags]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-trace_layers}
shift || true
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --no-cpu-baseline --no-extra-precisions --steps 6 --warmup 1 "$@" > $OUT/bench.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last 40 % of the run (timed steps, after autotune)
rows = rows[int(len(rows) * 0.6):]
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("void himo::", "").split("(")[0]
    agg[(name, r["Grid_Size_X"] + "x" + r.get("Grid_Size_Y", "1"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{name[:90]:90s} grid {grid:>12s}  n={len(v):4d}  avg {sum(v)/len(v):8.1f} us  share {100*sum(v)/tot:5.1f} %")
PY
