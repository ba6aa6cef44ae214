#!/bin/bash
# HBM traffic per launch from PMC counters: separate rocprofv3 passes for FETCH_SIZE and WRITE_SIZE
# (TCC slots: FETCH_SIZE needs 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md, rocprofv3 PMC slots).
# Run on the GPU box from the repo root:  bash scripts/collect_traffic.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in compdis pipeline fastnsf; do
  if [ $wl = fastnsf ]; then ARGS="--workload fastnsf --steps 1 --warmup 1 --fastnsf-iters 20 --no-cpu-baseline --single-stream"; elif [ $wl = compdis ]; then ARGS="--workload compdis --steps 3 --warmup 1 --no-cpu-baseline"; else ARGS="--workload pipeline --steps 2 --warmup 1 --no-cpu-baseline --no-extra-precisions --no-extra-workloads --single-stream --no-hostfed-leg"; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${wl}_$c -o pmc -- python $R/bench.py $ARGS > $OUT/${wl}_$c.log 2>&1
  done
done
cd $R
python scripts/parse_traffic.py $OUT > $OUT/traffic.json
cat $OUT/traffic.json
