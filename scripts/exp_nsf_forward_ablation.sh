#!/bin/bash
# FastNSF forward kernel, what is exposed?  Builds ABLATED copies of csrc/nsffused.hip (results are garbage, timings are the point) into
# build/variants/nsf_fwd_<v>/libhimo_amd.so:  nospill = the H_k spill stores removed; nolds = the A-operand LDS stores removed; nomfma = the
# matrix products removed; nomask = the ReLU-bit stores removed.  Then on the GPU box: python scripts/exp_nsf_forward_ablation.py
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
SRC=$R/himo_amd/csrc; make -C $SRC -j8 >/dev/null
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for v in nospill nolds nomfma nomask; do
  OUT=$R/build/variants/nsf_fwd_$v; mkdir -p $OUT
  cp $SRC/nsffused.hip $OUT/nsffused.hip
  case $v in
    nospill) python3 - $OUT/nsffused.hip <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
a = "                    *reinterpret_cast<uint4*>(sp + nsf_frag(wave, rt, j, 0)) = hi;\n                    *reinterpret_cast<uint4*>(sp + nsf_frag(wave, rt, j, 1)) = mid;\n"
assert a in s
s = s.replace(a, "                    bits ^= (hi.x & mid.y) == 0x12345678u ? 1u : 0u;      // (ablation: keep the split arithmetic alive)\n")
open(p, "w").write(s)
PY
    ;;
    nolds) python3 - $OUT/nsffused.hip <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
a = "                for (int r = 0; r < 16; ++r) nsf_a_store<false>(A, rt * 32 + nsf_row(r, lh), col, h[rt][r]);\n"
assert a in s
s = s.replace(a, "                for (int r = 0; r < 16; ++r) bits ^= h[rt][r] == 123.456f ? 1u : 0u;\n")
open(p, "w").write(s)
PY
    ;;
    nomfma) python3 - $OUT/nsffused.hip <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
a = "        nsf_fwd_gemm(A, a.w_hidden[k + 1], wave * 32, acc, li, lh);\n"
assert a in s
s = s.replace(a, "        acc[0][0] = A[threadIdx.x]; acc[1][0] = a.w_hidden[k + 1][threadIdx.x];\n")
open(p, "w").write(s)
PY
    ;;
    nomask) python3 - $OUT/nsffused.hip <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
a = "        a.maskbits[((int64_t)k * a.tiles + blockIdx.x) * 256 + threadIdx.x] = bits;\n"
assert a in s
s = s.replace(a, "        if (bits == 0x13572468u) a.maskbits[((int64_t)k * a.tiles + blockIdx.x) * 256 + threadIdx.x] = bits;\n")
open(p, "w").write(s)
PY
    ;;
  esac
  ( cd $OUT && $HIPCC -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I$SRC -c nsffused.hip -o nsffused.o ) &
done
wait
for v in nospill nolds nomfma nomask; do
  OUT=$R/build/variants/nsf_fwd_$v
  OBJS=""
  for f in $(cd $SRC && ls *.hip); do b=${f%.hip}; if [ $b = nsffused ]; then OBJS="$OBJS $OUT/nsffused.o"; else OBJS="$OBJS $R/build/csrc/$b.o"; fi; done
  $HIPCC --offload-arch=gfx950 -shared -fPIC -o $OUT/libhimo_amd.so $OBJS
  echo $OUT/libhimo_amd.so
done
