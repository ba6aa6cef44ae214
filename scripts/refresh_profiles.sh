#!/bin/bash
# Regenerates everything under profiles/ for one round (run on the MI355X box from the repo root, via gpurun):
#   bash scripts/refresh_profiles.sh r01
# Outputs land in gpurun_out/profiles_<round>/ (gpurun merges that back); copy them into profiles/ afterwards.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
RN=${1:-r01}
OUT=$R/gpurun_out/profiles_$RN
mkdir -p $OUT
cd $R
bash scripts/collect_traffic.sh > $OUT/traffic.log 2>&1            # PMC passes first: bench.py reads profiles/traffic_latest.json
cp gpurun_out/pmc/traffic.json $OUT/traffic_latest.json
cp $OUT/traffic_latest.json profiles/traffic_latest.json
python bench.py > $OUT/${RN}_bench_pipeline_n1.json 2> $OUT/pipeline.err
python bench.py --precision bf16x3 --no-cpu-baseline > $OUT/${RN}_bench_pipeline_n1_bf16x3.json 2>> $OUT/pipeline.err
python bench.py --float32-activations --no-cpu-baseline > $OUT/${RN}_bench_pipeline_n1_float32_activations.json 2>> $OUT/pipeline.err
python bench.py --workload compdis > $OUT/${RN}_bench_compdis_n1.json 2> $OUT/compdis.err
python bench.py --workload train > $OUT/${RN}_bench_train_n1.json 2> $OUT/train.err
python bench.py --workload train --train-batchnorm frozen --no-extra-workloads > $OUT/${RN}_bench_train_n1_frozen_bn.json 2>> $OUT/train.err
python bench.py --cloud rings --no-cpu-baseline > $OUT/${RN}_bench_pipeline_n1_lidar_rings.json 2>> $OUT/pipeline.err
python scripts/exp_layers.py 16 > $OUT/${RN}_conv3x3_per_layer.txt 2>&1
python scripts/exp_eval.py > $OUT/${RN}_evaluator_throughput.txt 2>&1
python scripts/exp_hostfed.py > $OUT/${RN}_hostfed_pipeline.txt 2>&1
python bench.py --workload fastnsf > $OUT/${RN}_bench_fastnsf_n1.json 2>> $OUT/pipeline.err
bash scripts/exp_clock_pmc.sh default > $OUT/${RN}_conv3x3_clock_and_mfma_busy.txt 2>&1
bash scripts/pmc_step_summary.sh > $OUT/${RN}_pmc_step_summary.txt 2>&1
python scripts/exp_savezip.py > $OUT/${RN}_exp_savezip.log 2>&1
python scripts/exp_h5_loader.py 4 12 > $OUT/${RN}_h5_loader.txt 2>&1
python scripts/exp_nn_grid.py > $OUT/${RN}_exp_nn_grid.txt 2>&1
python bench.py --workload train --cloud rings --no-extra-workloads --no-cpu-baseline > $OUT/${RN}_bench_train_n1_lidar_rings.json 2>> $OUT/train.err
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/mfma_peak scripts/micro/mfma_peak.hip && /tmp/mfma_peak > $OUT/${RN}_mfma_sustained_peak.txt 2>&1
cd /tmp && export TMPDIR=/tmp
python $R/scripts/exp_hbm_layers.py > $OUT/${RN}_hbm_side_layers.txt 2>&1
python $R/scripts/exp_upsample.py > $OUT/${RN}_upsample.txt 2>&1
for wl in pipeline compdis train train_rings fastnsf; do
  ARGS="--workload ${wl%_rings} --no-cpu-baseline --no-extra-precisions"
  [ $wl = train ] && ARGS="$ARGS --steps 5 --warmup 2 --no-extra-workloads"
  [ $wl = train_rings ] && ARGS="$ARGS --steps 5 --warmup 2 --no-extra-workloads --cloud rings"
  [ $wl = fastnsf ] && ARGS="$ARGS --steps 2 --warmup 1 --no-extra-workloads --single-stream"   # one fit at a time: a launch's duration is its own
  [ $wl = pipeline ] && ARGS="$ARGS --no-extra-workloads --single-stream --no-hostfed-leg"   # one batch in flight: a launch's duration is its own
  # (train: the weight gradients on the main stream, as in the region bench.py times its roofline kernel in -- with the side streams a launch's duration includes what co-runs)
  HIMO_TRAIN_SIDE_STREAM=$([ ${wl%_rings} = train ] && echo 0 || echo 1) timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$wl -o $wl -- python $R/bench.py $ARGS > $OUT/${RN}_bench_${wl}_n1_under_rocprof.json 2> $OUT/prof_$wl.err
  f=$(find $OUT/prof_$wl -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/${RN}_${wl}_rocprofv3_kernel_stats.csv
  rm -rf $OUT/prof_$wl
done
cd $R
ls -la $OUT
