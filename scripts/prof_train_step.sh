#!/bin/bash
# one steady-state training step per kernel (rocprofv3 kernel trace) -> gpurun_out/<name>.txt   usage: prof_train_step.sh <name> [bench args]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-train_step_kernels}; shift
OUT=$R/gpurun_out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tmp -o t -- python $R/bench.py --workload train --no-extra-workloads --steps 6 --warmup 3 "$@" > $OUT/$N.log 2>&1
F=$(find $OUT/prof_tmp -name "*kernel_trace.csv" | head -1)
python $R/scripts/prof_train_step.py $F ${TOP:-70} ${STEP_INDEX:-6} > $OUT/$N.txt
rm -rf $OUT/prof_tmp
cat $OUT/$N.txt
