#!/bin/bash
# idle time between consecutive kernels of the FastNSF fit (rocprofv3 kernel trace): what a captured graph could recover at most
# usage (GPU box, repo root): bash scripts/exp_nsf_gaps.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/nsf_gaps
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python $R/bench.py --workload fastnsf --no-cpu-baseline --no-extra-workloads --steps 3 --warmup 1 > $OUT/bench.log 2>&1
cd $R
python - <<PY
import csv, glob
rows = []
for f in glob.glob("$OUT/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows = [r for r in rows if "nsf_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gaps, busy = [], 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 200_000:                      # inside a fit (between frames the host prepares the next pair)
        gaps.append(g)
    busy += int(a["End_Timestamp"]) - int(a["Start_Timestamp"])
gaps.sort()
n = len(gaps)
print(f"{len(rows)} nsf_* dispatches; {n} gaps inside fits: median {gaps[n // 2] / 1e3:.2f} us, mean {sum(gaps) / n / 1e3:.2f} us, p90 {gaps[int(n * 0.9)] / 1e3:.2f} us")
print(f"kernel time {busy / 1e6:.2f} ms, gap time {sum(gaps) / 1e6:.2f} ms = {100 * sum(gaps) / (busy + sum(gaps)):.1f} % of the fits")
PY
rm -rf $OUT/*/ 
