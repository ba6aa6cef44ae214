#!/bin/bash
# per-kernel time of the training step: rocprofv3 kernel stats -> gpurun_out/<name>.csv  (usage: prof_train_stats.sh <name> [bench args])
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-train_kernel_stats}; shift
OUT=$R/gpurun_out
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tmp -o t -- python $R/bench.py --workload train --no-extra-workloads --steps 5 --warmup 2 "$@" > $OUT/$N.log 2>&1
find $OUT/prof_tmp -name "*kernel_stats.csv" -exec cp {} $OUT/$N.csv \;
rm -rf $OUT/prof_tmp
grep '^{' $OUT/$N.log | cut -c1-200
