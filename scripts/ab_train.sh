#!/bin/bash
# usage (GPU box): bash scripts/ab_train.sh <rounds> <variant> [<variant> ...]   ("default" = the in-tree library; others are
# build/variants/<name>/libhimo_amd.so): the training bench (uniform cloud, no extra legs) for every variant, interleaved `rounds`
# times on the SAME box; prints frames/s with and without the side streams and the 3x3 weight-gradient family's average launch.
cd ${GRAFT_REPO_ROOT:-.}
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset HIMO_AMD_LIB; else export HIMO_AMD_LIB=$PWD/build/variants/$v/libhimo_amd.so; fi
    python bench.py --workload train --no-extra-workloads --steps 30 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('%-10s train %7.2f frames/s  single stream %7.2f  wgrad launch %.4f ms' % ('$v', d['value'], d.get('value_single_stream') or 0, d['roofline']['avg_launch_ms']))"
  done
done
