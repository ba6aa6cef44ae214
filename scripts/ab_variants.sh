#!/bin/bash
# usage (GPU box): bash scripts/ab_variants.sh "<script + args>" <variant> [<variant> ...]   ("default" = the in-tree library)
cd ${GRAFT_REPO_ROOT:-.}
CMD=$1; shift
for v in "$@"; do
  echo "=== $v"
  if [ "$v" = default ]; then python $CMD; else HIMO_AMD_LIB=$PWD/build/variants/$v/libhimo_amd.so python $CMD; fi
done
