#!/bin/bash
# Builds a VARIANT of libhimo_amd.so with extra compile flags into build/variants/<name>/libhimo_amd.so (A/B experiments on the
# same box: HIMO_AMD_LIB=<that path> python bench.py ...).  usage: bash scripts/build_variant.sh <name> "<extra flags>" [file.hip ...]
# Only the listed translation units (default: all) get the extra flags; the rest reuse the default objects.
set -eu
NAME=$1; EXTRA=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/himo_amd/csrc; OUT=$R/build/variants/$NAME; mkdir -p $OUT
make -C $SRC -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FILES=${@:-$(cd $SRC && ls *.hip)}
OBJS=""
for f in $(cd $SRC && ls *.hip); do
  b=${f%.hip}
  if echo " $FILES " | grep -q " $f "; then
    fl=$(make -C $SRC -pn 2>/dev/null | sed -n "s/^FLAGS_$b := //p" | head -1)
    $HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $fl $EXTRA -c $SRC/$f -o $OUT/$b.o &
    OBJS="$OBJS $OUT/$b.o"
  else
    OBJS="$OBJS $R/build/csrc/$b.o"
  fi
done
wait
$HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libhimo_amd.so $OBJS
echo $OUT/libhimo_amd.so
