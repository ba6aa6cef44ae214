set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; RN=r05; OUT=$R/gpurun_out/profiles_final; mkdir -p $OUT; cd $R
python bench.py --workload train > $OUT/${RN}_bench_train_n1.json 2> $OUT/train.err
python bench.py --workload train --train-batchnorm frozen --no-extra-workloads > $OUT/${RN}_bench_train_n1_frozen_bn.json 2>> $OUT/train.err
python bench.py --workload train --cloud rings --no-extra-workloads --no-cpu-baseline > $OUT/${RN}_bench_train_n1_lidar_rings.json 2>> $OUT/train.err
python scripts/exp_train_batch.py 8 > $OUT/${RN}_exp_train_batch.txt 2>&1
python scripts/exp_linear_wgrad.py > $OUT/${RN}_exp_linear_wgrad.txt 2>&1
bash scripts/prof_train_step.sh train_step_final > /dev/null 2>&1; cp gpurun_out/train_step_final.txt $OUT/${RN}_train_step_kernels_and_streams.txt
cd /tmp && export TMPDIR=/tmp
for wl in train train_rings; do
  ARGS="--workload train --no-cpu-baseline --no-extra-precisions --steps 5 --warmup 2 --no-extra-workloads"
  [ $wl = train_rings ] && ARGS="$ARGS --cloud rings"
  HIMO_TRAIN_SIDE_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$wl -o $wl -- python $R/bench.py $ARGS > $OUT/${RN}_bench_${wl}_n1_under_rocprof.json 2> $OUT/prof_$wl.err
  f=$(find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${RN}_${wl}_rocprofv3_kernel_stats.csv; rm -rf $OUT/prof_$wl
done
ls $OUT
