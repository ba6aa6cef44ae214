#!/bin/bash
# usage (GPU box): bash scripts/ab_legs.sh  -- FastNSF and training frames/s of the in-tree library and of build/variants/{wg256,wg768}, twice (edit the list)
cd ${GRAFT_REPO_ROOT:-.}
for r in 1 2; do
for v in default wg256 wg768; do
  if [ "$v" = default ]; then unset HIMO_AMD_LIB; else export HIMO_AMD_LIB=$PWD/build/variants/$v/libhimo_amd.so; fi
  f=$(python bench.py --workload fastnsf --no-extra-workloads 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],3))")
  t=$(python bench.py --workload train --no-extra-workloads 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['value'],2))")
  echo "$v fastnsf $f train $t"
done
done
