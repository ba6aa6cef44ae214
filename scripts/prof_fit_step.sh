#!/bin/bash
# one steady-state optimiser step of the FED training program per kernel and stream (rocprofv3 kernel trace of seflow.fit.fit over .h5 scenes, batch_size 8,
# labels cached after the first epoch) -> gpurun_out/<name>.txt, to set beside r06_train_b8_step_kernels_and_streams.txt (the same step on resident samples)
# usage: prof_fit_step.sh <name> [step index]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; N=${1:-fit_step_kernels}; STEP=${2:-24}
OUT=$R/gpurun_out
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_tmp -o t -- python $R/scripts/prof_fit_step_run.py > $OUT/$N.log 2>&1
F=$(find $OUT/prof_tmp -name "*kernel_trace.csv" | head -1)
python $R/scripts/prof_train_step.py $F ${TOP:-40} $STEP > $OUT/$N.txt
tail -3 $OUT/$N.log >> $OUT/$N.txt
rm -rf $OUT/prof_tmp
cat $OUT/$N.txt
