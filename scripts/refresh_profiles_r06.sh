#!/bin/bash
# Regenerates the round-6 evidence under profiles/ (run on the MI355X box from the repo root, via gpurun):
#   bash scripts/refresh_profiles_r06.sh
# Outputs land in gpurun_out/profiles_r06/ (gpurun merges that back); copy them into profiles/ afterwards.
# (scripts/refresh_profiles.sh <round> is the full set of rounds 1-5; this one is what round 6 changed or added.)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
RN=r06
OUT=$R/gpurun_out/profiles_$RN
mkdir -p $OUT
cd $R
timeout 900 bash scripts/collect_traffic.sh > $OUT/traffic.log 2>&1            # PMC passes first: bench.py reads profiles/traffic_latest.json
cp gpurun_out/pmc/traffic.json $OUT/traffic_latest.json && cp $OUT/traffic_latest.json profiles/traffic_latest.json
timeout 1200 python bench.py > $OUT/${RN}_bench_default_n1.json 2> $OUT/default.err
timeout 300 python bench.py --workload train --no-extra-workloads > $OUT/${RN}_bench_train_n1.json 2> $OUT/train.err
timeout 300 python bench.py --workload train --train-batch 8 --no-extra-workloads > $OUT/${RN}_bench_train_b8_n1.json 2>> $OUT/train.err
timeout 300 python scripts/prof_fit_stages.py > $OUT/${RN}_fit_stages.txt 2>&1
timeout 300 python scripts/exp_eval.py > $OUT/${RN}_evaluator_throughput.txt 2>&1
timeout 300 python scripts/exp_eval_feed.py > $OUT/${RN}_exp_eval_feed.txt 2>&1
timeout 300 python scripts/exp_labels.py > $OUT/${RN}_exp_labels.txt 2>&1
timeout 300 python scripts/exp_fit_feed.py > $OUT/${RN}_exp_fit_feed.txt 2>&1
timeout 300 bash scripts/diag_legs.sh > $OUT/${RN}_legs_in_the_default_process.txt 2>&1
timeout 600 python scripts/exp_eval_main.py 257 2>&1 | grep -v "^ \|^$\|amdgpu.ids" > $OUT/${RN}_exp_eval_main_threads_vs_processes.txt
timeout 300 python scripts/exp_fork_cost.py > $OUT/${RN}_exp_fork_cost.txt 2>&1
timeout 300 python scripts/exp_train_host.py 8 > $OUT/${RN}_exp_train_host_b8.txt 2>&1
timeout 300 python scripts/exp_nsf_forward_ablation.py > $OUT/${RN}_exp_nsf_forward_ablation.txt 2>&1      # (after: bash scripts/exp_nsf_forward_ablation.sh on the build host)
timeout 900 bash scripts/prof_fit_step.sh ${RN}_fit_step_kernels_and_streams > /dev/null 2>&1; cp gpurun_out/${RN}_fit_step_kernels_and_streams.txt $OUT/
cd /tmp && export TMPDIR=/tmp
HIMO_EXP_LABELS_ONLY=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_labels -o lab -- python $R/scripts/exp_labels.py > /dev/null 2>&1
f=$(find $OUT/prof_labels -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${RN}_labels_rocprofv3_kernel_stats.csv; rm -rf $OUT/prof_labels
for wl in pipeline train train_b8 fastnsf; do
  ARGS="--workload ${wl%_b8} --no-cpu-baseline --no-extra-precisions"
  [ $wl = train ] && ARGS="$ARGS --steps 5 --warmup 2 --no-extra-workloads"
  [ $wl = train_b8 ] && ARGS="$ARGS --steps 5 --warmup 2 --no-extra-workloads --train-batch 8"
  [ $wl = fastnsf ] && ARGS="$ARGS --steps 2 --warmup 1 --no-extra-workloads --single-stream"   # one fit at a time: a launch's duration is its own
  [ $wl = pipeline ] && ARGS="$ARGS --no-extra-workloads --single-stream --no-hostfed-leg"   # one batch in flight: a launch's duration is its own
  # (train: the weight gradients on the main stream, as in the region bench.py times its roofline kernel in)
  HIMO_TRAIN_SIDE_STREAM=$([ ${wl%_b8} = train ] && echo 0 || echo 1) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$wl -o $wl -- python $R/bench.py $ARGS > $OUT/${RN}_bench_${wl}_n1_under_rocprof.json 2> $OUT/prof_$wl.err
  f=$(find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${RN}_${wl}_rocprofv3_kernel_stats.csv
  rm -rf $OUT/prof_$wl
done
cd $R
ls -la $OUT
