#!/bin/bash
# usage (GPU box): bash scripts/ab_bench.sh <rounds> <variant> [<variant> ...]   ("default" = the in-tree library): the default pipeline
# bench (no CPU leg, no extra legs) for every variant, interleaved `rounds` times on the SAME box; prints frames/s, the 3x3 family's
# average launch and the per-family ms of the profiled step.
cd ${GRAFT_REPO_ROOT:-.}
R=$1; shift
for r in $(seq 1 $R); do
  for v in "$@"; do
    if [ "$v" = default ]; then unset HIMO_AMD_LIB; else export HIMO_AMD_LIB=$PWD/build/variants/$v/libhimo_amd.so; fi
    python bench.py --no-cpu-baseline --no-extra-precisions --no-extra-workloads --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels']
print('%-10s %7.1f frames/s  conv3x3 launch %.4f ms  | ' % ('$v', d['value'], d['roofline']['avg_launch_ms']) + ' '.join('%s %.2f' % (n.replace('_kernel','').replace('_f16x2',''), v['ms_per_step']) for n, v in sorted(k.items(), key=lambda kv: -kv[1]['ms_per_step'])[:7]))"
  done
done
